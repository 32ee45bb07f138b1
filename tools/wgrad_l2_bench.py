"""A/B of the two "f16x2" weight-gradient kernels, one process, interleaved rounds:
   reg = fp32 operands split and transposed in registers (conv_h2.hip), dma = pre-split limb planes, LDS-DMA +
   transposed LDS reads (conv_l2.hip).  The split passes of the dma path are timed separately."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

SHAPES = [  # (B,H,W,Cin,Cout,k,dil)
    (4, 65, 129, 2048, 256, 3, 12), (4, 65, 129, 2048, 256, 3, 24), (4, 65, 129, 512, 512, 3, 4), (4, 65, 129, 256, 256, 3, 2),
    (4, 65, 129, 1280, 256, 3, 1), (4, 65, 129, 1024, 256, 1, 1), (4, 65, 129, 256, 1024, 1, 1), (4, 65, 129, 512, 2048, 1, 1),
    (4, 65, 129, 2048, 512, 1, 1), (4, 65, 129, 128, 128, 3, 1), (4, 129, 257, 64, 256, 1, 1), (4, 129, 257, 64, 64, 3, 1),
    (4, 129, 257, 256, 64, 1, 1), (4, 65, 129, 256, 32, 1, 1),
]
if os.environ.get("QUICK"):
    SHAPES = SHAPES[:5]
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
    dy = torch.randn(B, H, W, Cout, device="cuda")
    pad = dil * (k - 1) // 2

    def run(path):
        ops.H2_PATH = path
        return ops.conv_wgrad(x, dy, k, 1, dil, pad, Cout, Cin)

    r, d = run("reg"), run("dma")
    diff = ((r - d).norm() / r.norm()).item()
    t = {"reg": [], "dma": []}
    for _ in range(ROUNDS):
        t["reg"].append(timed(lambda: run("reg")))
        t["dma"].append(timed(lambda: run("dma")))
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    m = {kk: min(v) for kk, v in t.items()}
    sk = (ops._wgrad_splitk(B * H * W, Cout, Cin, k * k, False), ops._wgrad_splitk(B * H * W, Cout, Cin, k * k, True))
    print(f"Cin={Cin:5d} Cout={Cout:5d} k={k} d={dil:2d} M={B*H*W:6d} | reg {m['reg']*1e3:7.1f} us {fl/m['reg']/1e9:6.1f} TF | "
          f"dma {m['dma']*1e3:7.1f} us {fl/m['dma']/1e9:6.1f} TF (x{m['reg']/m['dma']:.2f}) | rel diff {diff:.2e} | splitk {sk}", flush=True)
