"""Which lines of this package issue device copies (aten.copy_ / clone / contiguous / _to_copy) during one adaptation step.

    python tools/copy_sites.py
"""
import collections
import os
import sys
import tempfile
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if any(k in str(func) for k in ("copy_", "clone", "_to_copy", "contiguous")):
            frames = [f for f in traceback.extract_stack() if "/onda_amd/" in f.filename or f.filename.endswith("bench.py")]
            where = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1])
            nbytes = 0
            for a in args:
                if torch.is_tensor(a):
                    nbytes = a.numel() * a.element_size()
                    break
            self.hits[(str(func), where, nbytes)] += 1
        return func(*args, **(kwargs or {}))


def main():
    sys.argv = sys.argv[:1]
    args = bench.parse()
    torch.cuda.set_device(0)
    if os.environ.get("EVAL"):  # the forward-only evaluation path (config 1) instead of the adaptation step
        return eval_path(args)
    with tempfile.TemporaryDirectory() as tmp:
        da, src, trg = bench.build_adapter(args, "cuda:0", tmp, 1)
        for i in range(3):
            bench.one_step(da, src, trg, i, 100)
        torch.cuda.synchronize()
        with Sites() as s:
            bench.one_step(da, src, trg, 3, 100)
        torch.cuda.synchronize()
        total = sum(s.hits.values())
        print(total, "copy-like dispatches in one step")
        agg = collections.Counter()
        for (fn, where, nb), n in s.hits.items():
            agg[(fn, where)] += n
        for (fn, where), n in agg.most_common(40):
            print(f"{n:5d}  {fn:28s} {where}")


def eval_path(args):
    from onda_amd import ops
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_model
    from onda_amd.synthetic import fill_state_dict, synth_batch
    cfg, _ = hybrid_switch_cfg(args.width, args.height, "cuda:0", "NONE", batch_size=1)
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, 3.0)
    model.eval()
    f = {k: v.to("cuda:0") for k, v in synth_batch(8, args.height, args.width, seed=4000).items()}
    hist = torch.zeros(19, 19, dtype=torch.int64, device="cuda:0")
    with torch.no_grad():
        for _ in range(3):
            ops.upsample_argmax_hist(model(f["image"])[1]["out"], f["label"], hist, 19)
        torch.cuda.synchronize()
        with Sites() as s:
            ops.upsample_argmax_hist(model(f["image"])[1]["out"], f["label"], hist, 19)
    torch.cuda.synchronize()
    print(sum(s.hits.values()), "copy-like dispatches in one forward of 8 frames")
    agg = collections.Counter()
    for (fn, where, nb), n in s.hits.items():
        agg[(fn, where, nb)] += n
    for (fn, where, nb), n in agg.most_common(40):
        print(f"{n:5d}  {fn:28s} {nb:10d} B  {where}")


if __name__ == "__main__":
    main()
