"""Reference point for DESIGN.md (not part of the product or of bench.py): the CPU oracle's
adaptation step executed by PyTorch-ROCm eager kernels (MIOpen convolutions) on the same GPU,
same synthetic 512x1024 bs=4 workload, fp32.  Usage: python tools/eager_baseline.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import model as omodel
from oracle.step import OracleAdapter
from onda_amd.synthetic import synth_batch, synth_prototypes, synth_tensor

dev = "cuda:0"
torch.backends.cudnn.benchmark = True
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, 1.0).to(dt).to(dev) for k, shape, dt in omodel.state_spec()}
ad = OracleAdapter(sd, tuple(t.to(dev) for t in synth_prototypes()))
ad.refresh_dynamic()
src = {k: v.to(dev) for k, v in synth_batch(4, 512, 1024, seed=1000).items()}
trg = {k: v.to(dev) for k, v in synth_batch(4, 512, 1024, seed=2000).items()}
# the oracle's distance loop allocates on the CPU; move its scratch to the GPU
import oracle.prototypes as op
_ones = torch.ones
op.torch.ones = lambda *a, **k: _ones(*a, **{**k, "device": dev})
def one():
    masks = tuple(omodel.draw_drop_mask(4, device=dev) for _ in range(3))
    ad.step(src, trg, masks)
    ad.update_ema()
for _ in range(2):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"PyTorch-ROCm eager (MIOpen, fp32): {dt*1e3:.1f} ms/step  {4/dt:.2f} images/s  branch={'dynamic' if ad.switch.current else 'static'}")
