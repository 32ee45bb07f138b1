"""Kernel-time probe of the pre-split forward kernel on chosen shapes (events around N back-to-back launches, inputs
pre-split; the environment knobs ONDA_CONV_SCHED / ONDA_L2_XT / ONDA_L2_NOSKIP are read once per process, so run one
process per variant).  STATS=0 drops the per-channel statistics from the epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

SHAPES = [  # (B,H,W,Cin,Cout,k,dil)
    (4, 65, 129, 1024, 256, 1, 1), (4, 64, 128, 1024, 256, 1, 1), (4, 65, 129, 256, 1024, 1, 1), (4, 64, 128, 256, 1024, 1, 1),
    (4, 65, 129, 512, 2048, 1, 1), (4, 65, 129, 2048, 512, 1, 1), (4, 65, 129, 256, 256, 3, 2), (4, 65, 129, 512, 512, 3, 4),
    (4, 65, 129, 128, 512, 1, 1), (4, 65, 129, 512, 128, 1, 1), (4, 129, 257, 64, 256, 1, 1), (4, 129, 257, 256, 64, 1, 1),
    (4, 65, 129, 128, 128, 3, 1), (4, 129, 257, 64, 64, 3, 1), (4, 129, 257, 64, 64, 1, 1), (4, 256, 512, 160, 64, 1, 1),
]
if os.environ.get("LONGK"):  # the long K loops (conv_l2_kernel<4,2>): layer3 / layer4 3x3 and an ASPP branch
    SHAPES = [(4, 65, 129, 256, 256, 3, 2), (4, 65, 129, 512, 512, 3, 4), (4, 65, 129, 2048, 256, 3, 12), (4, 65, 129, 2048, 512, 1, 1)]
if os.environ.get("QUICK6"):
    SHAPES = SHAPES[:6]
if os.environ.get("SMALL"):
    SHAPES = SHAPES[8:]
N = 20
stats = int(os.environ.get("STATS", "4"))
ops.H2_PATH = "dma"
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ONDA_") or k == "STATS")
print("##", tag or "default")
for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    ops.activation_limbs(x)
    pad = dil * (k - 1) // 2
    out = torch.empty(B, H, W, Cout, device="cuda")
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.conv_forward(x, wp, k, 1, dil, pad, Cout, out=out, want_stats=stats)
        e0.record()
        for _ in range(N):
            ops.conv_forward(x, wp, k, 1, dil, pad, Cout, out=out, want_stats=stats)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / N)
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    byt = B * H * W * (Cin * 4 + Cout * 4)
    print(f"Cin={Cin:5d} Cout={Cout:5d} k={k} d={dil:2d} M={B*H*W:6d} | {best*1e3:7.1f} us {fl/best/1e9:6.1f} TF  {byt/best/1e6:6.0f} GB/s(alg)", flush=True)
