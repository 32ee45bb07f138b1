"""Which host call sites launch the small device kernels of one adaptation step (fills, device-to-device copies,
elementwise torch kernels): torch.profiler with Python stacks over one step at the bench workload."""
import os, sys, tempfile, argparse, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
args = argparse.Namespace(gpus=1, steps=3, warmup=2, height=512, width=1024, batch=4, branch="dynamic",
                          no_cpu_baseline=True, no_roofline=True)
with tempfile.TemporaryDirectory() as tmp:
    da, src, trg = bench.build_adapter(args, "cuda:0", tmp)
    for i in range(3):
        bench.one_step(da, src, trg, i, 10)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        bench.one_step(da, src, trg, 3, 10)
        torch.cuda.synchronize()
    agg = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
            continue
        if ev.device_time_total <= 0 and not ev.kernels:
            continue
        site = "?"
        for fr in ev.stack:
            if "onda_amd" in fr or "bench.py" in fr:
                site = fr.strip()
                break
        agg[(ev.name, site[-110:])] += 1
    for (name, site), n in agg.most_common(60):
        print(f"{n:5d}  {name:28s} {site}")
