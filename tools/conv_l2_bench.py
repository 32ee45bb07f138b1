"""A/B of the two "f16x2" forward kernels on the dominant conv shapes, in ONE process, interleaved rounds:
   reg = activations split in-kernel (conv_h2.hip), dma = both operands pre-split, LDS-DMA only (conv_l2.hip).
Prints per shape: time of each kernel alone (the split pass of the dma path timed separately), TFLOP/s, and the
largest difference between the two results (bit-identical when neither launch is stream-K balanced)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

SHAPES = [  # (B,H,W,Cin,Cout,k,dil)
    (4, 65, 129, 2048, 256, 3, 12), (4, 65, 129, 256, 2048, 3, 12), (4, 65, 129, 2048, 256, 3, 24), (4, 65, 129, 256, 2048, 3, 24), (4, 65, 129, 512, 512, 3, 4),
    (4, 65, 129, 256, 256, 3, 2), (4, 65, 129, 1280, 256, 3, 1),
    (4, 65, 129, 1024, 256, 1, 1), (4, 65, 129, 256, 1024, 1, 1), (4, 65, 129, 512, 2048, 1, 1), (4, 65, 129, 2048, 512, 1, 1),
    (4, 65, 129, 128, 128, 3, 1), (4, 65, 129, 128, 512, 1, 1), (4, 129, 257, 64, 256, 1, 1), (4, 129, 257, 64, 64, 3, 1),
    (4, 129, 257, 256, 64, 1, 1),
]
if os.environ.get("QUICK"):
    SHAPES = SHAPES[:4]
ROUNDS, N = 3, 5


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N


for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    pad = dil * (k - 1) // 2
    res = {}

    def run(path):
        ops.H2_PATH = path
        return ops.conv_forward(x, wp, k, 1, dil, pad, Cout, want_stats=True)

    def split():
        x._onda_limbs = None
        ops.activation_limbs(x)

    ys = {}
    for path in ("reg", "dma"):
        y, st, _ = run(path)
        ys[path] = (y.clone(), st.sum(0).clone())
    diff = (ys["reg"][0] - ys["dma"][0]).abs().max().item()
    sdiff = ((ys["reg"][1] - ys["dma"][1]).abs().max() / ys["reg"][1].abs().max()).item()
    t = {"reg": [], "dma": [], "split": []}
    for _ in range(ROUNDS):
        t["reg"].append(timed(lambda: run("reg")))
        t["dma"].append(timed(lambda: run("dma")))
        t["split"].append(timed(split))
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    m = {kk: min(v) for kk, v in t.items()}
    print(f"Cin={Cin:5d} Cout={Cout:5d} k={k} d={dil:2d} M={B*H*W:6d} | reg {m['reg']*1e3:7.1f} us {fl/m['reg']/1e9:6.1f} TF | "
          f"dma {m['dma']*1e3:7.1f} us {fl/m['dma']/1e9:6.1f} TF (x{m['reg']/m['dma']:.2f}) | split {m['split']*1e3:6.1f} us | "
          f"maxdiff {diff:.2e} stats {sdiff:.1e} | variant {ops.query('onda_conv_l2_variant', B*H*W, Cout)}", flush=True)
