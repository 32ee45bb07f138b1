"""Where the HOST time of one adaptation step goes (cProfile over 3 steps at the bench workload)."""
import cProfile, pstats, os, sys, tempfile, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
args = argparse.Namespace(gpus=1, steps=3, warmup=2, height=512, width=1024, batch=4, branch="dynamic",
                          no_cpu_baseline=True, no_roofline=True)
with tempfile.TemporaryDirectory() as tmp:
    da, src, trg = bench.build_adapter(args, "cuda:0", tmp)
    for i in range(2):
        bench.one_step(da, src, trg, i, 10)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for i in range(3):
        bench.one_step(da, src, trg, 2 + i, 10)
    pr.disable()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"issue time per step {t_issue/3*1e3:.1f} ms, wall per step {t_all/3*1e3:.1f} ms (with cProfile overhead)")
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.print_callers("method 'to' of")
    st.print_callers("Event.synchronize")
