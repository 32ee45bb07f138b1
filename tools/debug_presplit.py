import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops
torch.manual_seed(0)
for (Cin, Cout, k, H, W) in [(64, 256, 1, 17, 33), (256, 256, 1, 17, 33), (512, 256, 1, 17, 33), (64, 256, 1, 65, 129), (256, 256, 1, 65, 129), (32, 256, 1, 17, 33), (96, 256, 1, 17, 33)]:
    x = torch.randn(2, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    ops.PRESPLIT = True
    y1 = ops.conv_forward(x, wp, k, 1, 1, 0, Cout)[0].clone()
    ops.PRESPLIT = False
    y0 = ops.conv_forward(x, wp, k, 1, 1, 0, Cout)[0].clone()
    ref = torch.einsum("bhwc,oc->bhwo", x.double(), w[:, :, 0, 0].double()).float()
    e1 = (y1 - ref).abs().view(-1, Cout); e0 = (y0 - ref).abs().view(-1, Cout)
    bad = (e1 > 1e-4).nonzero()
    print(Cin, Cout, H, W, "presplit max err", e1.max().item(), "old", e0.max().item(), "bad count", len(bad))
    if len(bad):
        rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
        print("  bad rows", rows[:20].tolist(), "... n", len(rows), " bad cols n", len(cols), cols[:16].tolist())
