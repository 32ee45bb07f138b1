"""Split-K sweep of the pre-split weight-gradient path (kernel + slab reduction) per shape of the network: the time of every
split count around the heuristic's choice (ops.conv._wgrad_splitk), one process, interleaved repeats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

BATCH = int(os.environ.get("SWEEP_BATCH", "4"))  # 8: the paired student pass
SHAPES = [  # (B,H,W,Cin,Cout,k,dil)
    (4, 65, 129, 512, 512, 3, 4), (4, 65, 129, 2048, 256, 3, 12), (4, 65, 129, 256, 256, 3, 2), (4, 65, 129, 1280, 256, 3, 1),
    (4, 65, 129, 2048, 512, 1, 1), (4, 65, 129, 512, 2048, 1, 1), (4, 65, 129, 1024, 2048, 1, 1), (4, 65, 129, 1024, 256, 1, 1),
    (4, 65, 129, 256, 1024, 1, 1), (4, 65, 129, 512, 1024, 1, 1), (4, 65, 129, 1024, 512, 1, 1), (4, 65, 129, 128, 128, 3, 1),
    (4, 65, 129, 128, 512, 1, 1), (4, 65, 129, 512, 128, 1, 1), (4, 129, 257, 64, 64, 3, 1), (4, 129, 257, 64, 256, 1, 1),
    (4, 129, 257, 256, 64, 1, 1),
]
N = 8
ops.H2_PATH = "dma"
heur = ops.conv._wgrad_splitk
for (_b, H, W, Cin, Cout, k, dil) in SHAPES:
    B = BATCH
    x = torch.randn(B, H, W, Cin, device="cuda")
    dy = torch.randn(B, H, W, Cout, device="cuda")
    pad = dil * (k - 1) // 2
    xl, dyl = ops.limbs_of(x), ops.limbs_of(dy)
    M = B * H * W
    sk0 = heur(M, Cout, Cin, k * k, True)
    cands = sorted({max(1, int(round(sk0 * f))) for f in (0.5, 0.75, 1.0, 1.25, 1.5, 2.0)} | {max(1, sk0 - 1), sk0 + 1})
    res = {}
    for rep in range(3):
        for sk in cands:
            ops.conv._wgrad_splitk = lambda *a, _sk=sk, **kw: _sk
            try:
                ops.conv_wgrad(x, dy, k, 1, dil, pad, Cout, Cin)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(N):
                    ops.conv_wgrad(x, dy, k, 1, dil, pad, Cout, Cin)
                e1.record()
                torch.cuda.synchronize()
                res[sk] = min(res.get(sk, 1e9), e0.elapsed_time(e1) / N * 1e3)
            except Exception as ex:
                res[sk] = float("nan")
    best = min(res, key=lambda s: res[s] if res[s] == res[s] else 1e9)
    fl = 2.0 * M * Cout * Cin * k * k
    print(f"Cin={Cin:5d} Cout={Cout:5d} k={k} M={M:6d} | heuristic sk={sk0:3d} {res[sk0]:7.1f} us ({fl / res[sk0] / 1e6:5.0f} TF) | best sk={best:3d} {res[best]:7.1f} us"
          f" | " + " ".join(f"{s}:{res[s]:.0f}" for s in cands), flush=True)
ops.conv._wgrad_splitk = heur
