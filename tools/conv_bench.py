"""Time selected conv shapes (forward kernel) under the scheduling override ONDA_CONV_SCHED."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops
SHAPES = [  # (B,H,W,Cin,Cout,k,dil)
    (4, 65, 129, 256, 256, 3, 2), (4, 65, 129, 1024, 256, 1, 1), (4, 65, 129, 256, 1024, 1, 1),
    (4, 65, 129, 2048, 256, 3, 12), (4, 65, 129, 256, 2048, 3, 12), (4, 65, 129, 512, 512, 3, 4),
    (4, 65, 129, 512, 2048, 1, 1), (4, 65, 129, 2048, 512, 1, 1), (4, 65, 129, 128, 128, 3, 1), (4, 129, 257, 64, 256, 1, 1),
]
for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    if os.environ.get("ZERO"):  # all-zero activations: same instruction stream and bytes, less switching power
        x.zero_()
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    pad = dil * (k - 1) // 2
    for _ in range(3):
        ops.conv_forward(x, wp, k, 1, dil, pad, Cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.conv_forward(x, wp, k, 1, dil, pad, Cout)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    print(f"sched={os.environ.get('ONDA_CONV_SCHED','auto'):5s} Cin={Cin:5d} Cout={Cout:5d} k={k} d={dil:2d} M={B*H*W:6d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
