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


if __name__ == "__main__":
    main()
