"""In-kernel timeline of conv_l2a_kernel (ONDA_L2X_STAMP=1): per workgroup, s_memtime at the start, then per work item
(row panel, column tile): rows ready (panel load on a panel change), end of the K loop, end of the epilogue.  Prints the mean
over the workgroups per item position, in ticks and in nanoseconds by the measured tick rate."""
import os, sys
os.environ["ONDA_L2X_STAMP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

SHAPES = [(4, 65, 129, 256, 1024), (4, 64, 128, 256, 1024), (4, 65, 129, 128, 512), (4, 129, 257, 64, 256)]
stats = int(os.environ.get("STATS", "4"))
for (B, H, W, Cin, Cout) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") / Cin ** 0.5
    wp = ops.pack_weight_fwd(w)
    ops.activation_limbs(x)
    out = torch.empty(B, H, W, Cout, device="cuda")
    ws = ops._conv_ws(x.device)
    for _ in range(3):
        ops.conv_forward(x, wp, 1, 1, 1, 0, Cout, out=out, want_stats=stats)
    torch.cuda.synchronize()
    ws.view(torch.int64)[-1024 * 32:].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_forward(x, wp, 1, 1, 1, 0, Cout, out=out, want_stats=stats)
    e1.record()
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[-1024 * 32:].reshape(1024, 32).cpu()
    st = st[st[:, 0] > 0]
    span = st.max().item() - st[:, 0].min().item()
    us = e0.elapsed_time(e1) * 1e3
    ns_per_tick = us * 1e3 / span
    print(f"Cin={Cin} Cout={Cout} M={B*H*W}: {st.shape[0]} workgroups, event {us:.1f} us, span {span} ticks, {ns_per_tick:.2f} ns/tick")
    ks = st[:, 16:24]
    if (ks > 0).all():
        d = (ks[:, 1:] - ks[:, :-1]).float().mean(0).tolist()
        print("   third item, K-step to K-step (ticks, stamped in front of each step's epilogue slice):", " ".join(f"{x:6.0f}" for x in d))
    st = st[:, :16]
    n = int((st > 0).sum(1).min().item())
    items = (n - 1) // 2
    for q in range(items):
        seg = st[:, 1 + 2 * q:3 + 2 * q] - st[:, 2 * q:2 + 2 * q]
        m = seg.float().mean(0).tolist()
        print(f"   item {q}: rows ready {m[0]:7.0f}  K loop (with the previous tile's epilogue inside) {m[1]:7.0f} ticks")
