"""In-kernel timeline of conv_l2x_kernel (ONDA_L2X_STAMP=1): per workgroup, s_memtime at the start, at the end of every
work item's K loop and at the end of its epilogue.  Prints, per work item position, the mean K-loop and epilogue
time over the workgroups (in s_memtime ticks and in microseconds by the measured tick rate)."""
import os, sys
os.environ["ONDA_L2X_STAMP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

SHAPES = [(4, 65, 129, 1024, 256, 1, 1), (4, 64, 128, 1024, 256, 1, 1), (4, 65, 129, 256, 1024, 1, 1), (4, 64, 128, 256, 1024, 1, 1),
          (4, 65, 129, 512, 2048, 1, 1), (4, 65, 129, 128, 512, 1, 1)]
ops.H2_PATH = "dma"
for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    ops.activation_limbs(x)
    out = torch.empty(B, H, W, Cout, device="cuda")
    ws = ops._conv_ws(x.device)
    for _ in range(3):
        ops.conv_forward(x, wp, k, 1, dil, 0, Cout, out=out, want_stats=4)
    torch.cuda.synchronize()
    ws.view(torch.int64)[-1024 * 32:].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_forward(x, wp, k, 1, dil, 0, Cout, out=out, want_stats=4)
    e1.record()
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[-1024 * 32:].reshape(1024, 32).cpu()
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min().item()
    ep = st[:, 24:29]
    st = st[:, :24]
    n = int((st > 0).sum(1).max().item())
    d = (ep[:, 1:] - ep[:, :-1]).float().mean(0).tolist()
    print("   last tile's epilogue phases (mean ticks): stats %.0f | transpose+stores %.0f | to barrier %.0f | stats reduce+store %.0f" % tuple(d))
    span = (st.max().item() - t0)
    us = e0.elapsed_time(e1) * 1e3
    print(f"Cin={Cin} Cout={Cout} M={B*H*W}: {st.shape[0]} workgroups, event time {us:.1f} us, kernel span {span} ticks "
          f"(~{span/us:.0f} ticks/us if the launch were the whole event time)")
    print("   start skew (ticks): mean %.0f max %d" % ((st[:, 0] - t0).float().mean().item(), (st[:, 0] - t0).max().item()))
    prev = st[:, 0]
    for i in range(1, n):
        cur = st[:, i]
        ok = cur > 0
        d = (cur - prev)[ok].float()
        kind = "K loop  " if i % 2 == 1 else "epilogue"
        print(f"   item {(i-1)//2} {kind}: n={int(ok.sum())} mean {d.mean().item():8.0f} min {d.min().item():8.0f} max {d.max().item():8.0f} ticks;"
              f" end at mean {(cur[ok]-t0).float().mean().item():9.0f}")
        prev = torch.where(ok, cur, prev)
