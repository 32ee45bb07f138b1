"""How far the host runs ahead of the device in the bench's adaptation step: host time to ENQUEUE each step (no
synchronisation) against the device time per step, and where inside a step the host spends its time.

    python tools/host_ahead.py [steps]
"""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    sys.argv = sys.argv[:1]
    args = bench.parse()
    torch.cuda.set_device(0)
    with tempfile.TemporaryDirectory() as tmp:
        da, src, trg = bench.build_adapter(args, "cuda:0", tmp, 1)
        for i in range(3):
            bench.one_step(da, src, trg, i, 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        host, marks, done = [], [], []
        for i in range(steps):
            a = time.perf_counter()
            bench.one_step(da, src, trg, 3 + i, 100)
            host.append(time.perf_counter() - a)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
            done.append(time.perf_counter() - t0)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{steps} steps: host enqueue {t_enq / steps * 1e3:.1f} ms per step, device {t_all / steps * 1e3:.1f} ms per step")
        print("host ms per step:", [round(h * 1e3, 1) for h in host])
        # how far behind the host the device finishes each step (ms): the host's lead when it leaves the step
        print("host lead at the end of each step (ms):", [round(start.elapsed_time(ev) - d * 1e3, 1) for ev, d in zip(marks, done)])
        if os.environ.get("ONDA_HOST_AHEAD_SHORT"):
            return
        # where the host time goes: cProfile of two steps, device kept busy
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(2):
            bench.one_step(da, src, trg, 20 + i, 100)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
        # ... and where it WAITS once the device is the slower side: eight more steps, by own time
        for i in range(4):
            bench.one_step(da, src, trg, 30 + i, 100)
        pr = cProfile.Profile()
        pr.enable()
        for i in range(8):
            bench.one_step(da, src, trg, 40 + i, 100)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
