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
for name, flops, e0, e1, tag, issued in ops.profile_entries(ops.PROFILE):
    a = agg.setdefault((name, tag), [0.0, 0.0, 0, 0.0])
    a[0] += flops; a[1] += e0.elapsed_time(e1); a[2] += 1; a[3] += issued
tot = sum(a[1] for a in agg.values())
print(f"total conv ms {tot:.2f}  flops {sum(a[0] for a in agg.values())/1e12:.2f} T")
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
# per shape: the roof that binds it = min(MFMA roof, arithmetic intensity x achievable HBM rate).  MFMA roof: f16 dense peak
# / 3 limb products (833 TF/s of fp32-equivalent flops); HBM: 6.3 TB/s (what a copy achieves, MI355X_MICROARCH.md), bytes =
# operands + output touched once (limb planes are 4 B per element, like fp32; weight-gradient slabs written once).
MFMA_ROOF, HBM_TBS = 2500.0 / 3, 6.3
def alg_bytes(tag):
    kind, M, co, ci, k = tag[0], tag[1], tag[2], tag[3], tag[4]
    if kind == "wgrad":
        return 4.0 * (M * ci + M * co) + 4.0 * tag[7] * co * k * k * ci
    return 4.0 * (M * ci + co * k * k * ci + M * co)
print("      ms  share   n     TF/s   roof(bound)  frac  issued  kernel                       (kind, M, Cout, Cin, k, stride, dil[, splitk])")
for (name, tag), (fl, ms, n, iss) in rows:
    ai = fl / n / alg_bytes(tag)                     # flop per byte of one launch
    hbm_roof = ai * HBM_TBS                          # TF/s (flop/B x TB/s)
    roof, bound = (MFMA_ROOF, "mfma") if MFMA_ROOF <= hbm_roof else (hbm_roof, "hbm")
    tf = fl / ms / 1e9
    print(f"{ms:8.3f} ms {100*ms/tot:5.1f}%  n={n:2d}  {tf:7.1f}  {roof:6.0f}({bound:4s})  {tf/roof:5.2f}  {iss/fl:5.3f}  {name:28s} {tag}")
