"""Per-shape conv timing on the GPU: one train-mode forward+backward of the model at the
BASELINE size with event instrumentation; prints TFLOP/s per (kind, M, N, K, k, stride, dil)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops
from onda_amd.framework.model.deeplabv2 import get_deeplab_v2
from onda_amd.synthetic import fill_state_dict, synth_batch

B, H, W = int(os.environ.get("B", 4)), int(os.environ.get("H", 512)), int(os.environ.get("W", 1024))
m = get_deeplab_v2(19, True, [3, 4, 6, 3], "ProDA"); m.multi_level = False
fill_state_dict(m, 1, 3.0); m = m.to("cuda:0").train()
b = synth_batch(B, H, W)
x, lab = b["image"].cuda(), b["label_res"].cuda()
for it in range(3):
    ops.PROFILE = [] if it == 2 else None
    _, o = m(x)
    loss = ops.seg_losses(o["out"], lab, 1.0, 0.0, 0.0)[0]
    loss.backward()
    m.zero_grad()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, flops, e0, e1, tag in ops.profile_entries(ops.PROFILE):
    a = agg.setdefault((name, tag), [0.0, 0.0, 0])
    a[0] += flops; a[1] += e0.elapsed_time(e1); a[2] += 1
tot = sum(a[1] for a in agg.values())
print(f"total conv ms {tot:.2f}  flops {sum(a[0] for a in agg.values())/1e12:.2f} T")
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (name, tag), (fl, ms, n) in rows:
    print(f"{ms:8.3f} ms {100*ms/tot:5.1f}%  n={n:2d}  {fl/ms/1e9:7.1f} TF/s  {name:28s} {tag}")
