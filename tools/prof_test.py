import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
t0 = time.time()
import test_hip_model as T
from conftest import load_golden
cache = {}
def golden(name):
    if name not in cache: cache[name] = load_golden(name)
    return cache[name]
print("import", time.time() - t0, flush=True)
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
T.test_train_forward_backward_golden(golden, False)
print("train fwd test", time.time() - t0, flush=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
