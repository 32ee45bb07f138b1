"""Which Python call sites create zero-filled device tensors during one adaptation step (torch.zeros / zeros_like /
Tensor.zero_ / new_zeros wrapped with a counter)."""
import os, sys, tempfile, argparse, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
args = argparse.Namespace(gpus=1, steps=3, warmup=2, height=512, width=1024, batch=4, branch="dynamic",
                          no_cpu_baseline=True, no_roofline=True)
sites = collections.Counter()
active = [False]


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if active[0]:
            fr = [x for x in traceback.extract_stack(limit=8)[:-1] if "onda_amd" in x.filename or "bench.py" in x.filename]
            key = f"{name} <- " + (f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno}" if fr else "torch internals")
            sites[key] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for owner, name in ((torch, "zeros"), (torch, "zeros_like"), (torch.Tensor, "zero_"), (torch.Tensor, "new_zeros"), (torch, "full"),
                    (torch.Tensor, "fill_"), (torch, "ones"), (torch, "ones_like")):
    wrap(owner, name)
with tempfile.TemporaryDirectory() as tmp:
    da, src, trg = bench.build_adapter(args, "cuda:0", tmp)
    for i in range(3):
        bench.one_step(da, src, trg, i, 10)
    torch.cuda.synchronize()
    active[0] = True
    bench.one_step(da, src, trg, 3, 10)
    torch.cuda.synchronize()
    active[0] = False
for k, v in sites.most_common(25):
    print(f"{v:5d}  {k}")
