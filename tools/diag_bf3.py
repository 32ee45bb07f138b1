import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from onda_amd import ops
from onda_amd.framework.model import deeplabv2
from onda_amd.framework.model.deeplabv2 import get_deeplab_v2
from onda_amd.synthetic import fill_state_dict, synth_batch
from conftest import load_golden
g = load_golden("g2_train_small")
mask = torch.from_numpy(g["drop_mask"])
deeplabv2.drop_mask_fn = lambda B, C, p, dev: mask.to(dev)
b = synth_batch(2, 64, 128, seed=7)
def run(mode):
    ops.CONV_MODE = mode
    m = get_deeplab_v2(19, True, [3,4,6,3], "ProDA"); m.multi_level = False
    fill_state_dict(m, 1, 3.0); m = m.cuda().train()
    _, o = m(b["image"].cuda())
    loss = ops.seg_losses(o["out"], b["label_res"].cuda(), 1.0, 0.0, 0.0)[0]
    loss.backward()
    return o["out"].detach().double().cpu(), {n: p.grad.double().cpu() for n, p in m.named_parameters() if p.grad is not None}
o1, g1 = run("f32"); o2, g2 = run("bf16x3")
print("out rel", ((o1-o2).abs().max()/o1.abs().max()).item())
rows = sorted(((( (g1[n]-g2[n]).norm()/g1[n].norm()).item(), n) for n in g1), reverse=True)
for e, n in rows[:12]: print(f"{e:.3e} {n}")
print("median", np.median([e for e,_ in rows]))
# unit-level: conv dgrad / fwd with scaled data
torch.manual_seed(0)
for scale_x, scale_g in ((1.0, 1.0), (1.0, 1e-7), (30.0, 1e-7), (1e-3, 1e-9)):
    x = (torch.randn(2, 9, 17, 512, device="cuda") * scale_x).requires_grad_(True)
    w = (torch.randn(2048, 512, 1, 1, device="cuda") / 22).requires_grad_(True)
    res = {}
    for mode in ("f32", "bf16x3"):
        ops.CONV_MODE = mode
        x.grad = None; w.grad = None
        y, _ = ops.Conv2dFn.apply(x, w, None, ops._PackCache(), 1, 1, 0, False, None)
        gy = torch.randn(2, 9, 17, 2048, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * scale_g
        y.backward(gy)
        res[mode] = (y.detach().double(), x.grad.double(), w.grad.double())
    print(scale_x, scale_g, [((a-b).norm()/a.norm()).item() for a, b in zip(res["f32"], res["bf16x3"])])
