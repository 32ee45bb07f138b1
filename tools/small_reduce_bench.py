"""Per-launch time of the column reductions around the ASPP head (GroupNorm statistics, SE pooling, bias gradients) at the
BASELINE sizes: tools/small_reduce_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops
from onda_amd._lib import call, query

def timeit(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B in (4, 8):
    H, W = 65, 129
    x = torch.randn(B, H, W, 256, device="cuda")
    cat = torch.randn(B, H, W, 1280, device="cuda")
    sl = cat[..., 256:512]
    print(f"B={B}: colsum [N,256] {timeit(lambda: ops.colsum(x)):6.1f} us | per image {timeit(lambda: ops.colsum(x, per_image=True)):6.1f} | "
          f"slice of the concat {timeit(lambda: ops.colsum(sl, per_image=True)):6.1f} | SE pool [N,1280] {timeit(lambda: ops.colsum(cat, alpha=1.0 / (H * W), per_image=True)):6.1f} us "
          f"({cat.numel() * 4 / 1e6:.0f} MB)", flush=True)
