"""Census of live CUDA tensors (by shape and dtype) after 3 and after 13 adaptation steps: what a step leaves behind."""
import os, sys, gc, tempfile, argparse, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
args = argparse.Namespace(gpus=1, steps=3, warmup=2, height=512, width=1024, batch=4, branch="dynamic", no_cpu_baseline=True, no_roofline=True)


def census():
    gc.collect()
    torch.cuda.synchronize()
    c = collections.Counter()
    seen = set()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                key = o.untyped_storage().data_ptr()
                if key in seen:
                    continue
                seen.add(key)
                c[(tuple(o.shape), str(o.dtype), o.untyped_storage().nbytes())] += 1
        except Exception:
            pass
    return c


with tempfile.TemporaryDirectory() as tmp:
    da, src, trg = bench.build_adapter(args, "cuda:0", tmp)
    for i in range(3):
        log = bench.one_step(da, src, trg, i, 100)
    a = census(); m0 = torch.cuda.memory_allocated()
    for i in range(10):
        log = bench.one_step(da, src, trg, 3 + i, 100)
    b = census(); m1 = torch.cuda.memory_allocated()
    print("allocated %.2f -> %.2f GB over 10 steps" % (m0 / 2**30, m1 / 2**30))
    diff = {k: b[k] - a.get(k, 0) for k in b if b[k] != a.get(k, 0)}
    for k, v in sorted(diff.items(), key=lambda kv: -kv[0][2] * kv[1])[:25]:
        print(f"{v:+4d} x {k[2]/2**20:9.2f} MB  {k[0]} {k[1]}")
