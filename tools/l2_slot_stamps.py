"""Where one K-step of the long-K kernel goes (measurement build -DONDA_L2_ABLATIONS, ONDA_L2_DBG=12): s_memtime stamps of lane 0
of the first wave of each half of every workgroup, at the K-step in the middle of its first whole tile:
  [0] slot start  [1] after the early half's vmcnt wait  [2] after the barrier  [3] after the DMA issue  [4] after the fragment reads
  have RETURNED (an explicit lgkmcnt(0) that the shipped kernel does not have)  [5] after the late half's vmcnt wait  [6] after the
  second barrier  [7] after the 48 MFMAs have been ISSUED.
usage: ONDA_LIB_PATH=<ablation lib> ONDA_L2_DBG=12 python tools/l2_slot_stamps.py"""
import os, sys
os.environ["ONDA_L2X_STAMP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

for (B, H, W, Cin, Cout, k, dil) in [(4, 65, 129, 512, 512, 3, 4), (4, 65, 129, 2048, 256, 3, 12)]:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    ops.activation_limbs(x)
    out = torch.empty(B, H, W, Cout, device="cuda")
    ws = ops._conv_ws(x.device)
    pad = dil * (k - 1) // 2
    for _ in range(3):
        ops.conv_forward(x, wp, k, 1, dil, pad, Cout, out=out)
    torch.cuda.synchronize()
    ws.view(torch.int64)[-1024 * 32:].zero_()
    ops.conv_forward(x, wp, k, 1, dil, pad, Cout, out=out)
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[-1024 * 32:][: 256 * 2 * 8].reshape(256, 2, 8).cpu().double()
    names = ["vmcnt wait (early)", "barrier", "DMA issue", "reads issued + returned", "vmcnt wait (late)", "barrier", "48 MFMAs issued"]
    print(f"Cin={Cin} Cout={Cout} k={k} d={dil}:")
    for half, label in ((0, "early half"), (1, "late half ")):
        rows = st[:, half]
        rows = rows[(rows > 0).all(1) & ((rows[:, 1:] - rows[:, :-1]) >= 0).all(1)]  # (workgroups whose first whole tile was stamped in full)
        d = (rows[:, 1:] - rows[:, :-1]).mean(0)
        print(f"  {label} ({rows.shape[0]} workgroups): " + " | ".join(f"{n} {v:.0f}" for n, v in zip(names, d.tolist())) + f" | total {float((rows[:, 7] - rows[:, 0]).mean()):.0f} ticks")
