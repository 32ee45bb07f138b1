"""Soak: N consecutive adaptation steps at the bench workload (every returned log kept, as a training loop that records
its history would): ms per step in blocks of 100, allocated / peak / reserved memory after every block, the target loss.
usage: python tools/soak.py [steps=500] [dynamic|static]"""
import os, sys, time, tempfile, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
args = argparse.Namespace(gpus=1, steps=3, warmup=2, height=512, width=1024, batch=4, branch=sys.argv[2] if len(sys.argv) > 2 else "dynamic", no_cpu_baseline=True,
                          no_roofline=True)
with tempfile.TemporaryDirectory() as tmp:
    da, src, trg = bench.build_adapter(args, "cuda:0", tmp)
    logs = []
    for i in range(3):
        bench.one_step(da, src, trg, i, steps + 3)
    torch.cuda.synchronize()
    t0, first = time.perf_counter(), None
    for i in range(steps):
        logs.append(bench.one_step(da, src, trg, 3 + i, steps + 3))
        if (i + 1) % 100 == 0 or i + 1 == steps:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n = (i % 100) + 1
            loss = float(logs[-1]["Total target loss"])
            first = loss if first is None else first
            print(f"steps {i + 1 - n:4d}-{i + 1:4d}: {1e3 * (t1 - t0) / n:7.2f} ms/step  allocated {torch.cuda.memory_allocated() / 2**30:5.2f} GB  "
                  f"peak {torch.cuda.max_memory_allocated() / 2**30:5.2f} GB  reserved {torch.cuda.memory_reserved() / 2**30:5.2f} GB  "
                  f"target loss {loss:.4f}", flush=True)
            t0 = time.perf_counter()
    print("switch:", da.model_select.current, "(1 = dynamic)")
    bad = [i for i, l in enumerate(logs) if not torch.isfinite(torch.as_tensor(float(l["Total target loss"])))]
    print("non-finite losses:", bad[:5] if bad else "none")
