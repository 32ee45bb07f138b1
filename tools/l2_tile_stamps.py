"""Where a tile of conv_l2_kernel<4,2> spends its time (measurement build -DONDA_L2_ABLATIONS, ONDA_L2_DBG=5): per workgroup the
s_memtime ticks of set-up (row decomposition, first DMAs), K loop and epilogue, summed over its work items, and its total.
    ONDA_LIB_PATH=tools/_ab_dbg/libonda_hip.so ONDA_L2_DBG=5 python tools/l2_tile_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops

SHAPES = [(4, 65, 129, 256, 256, 3, 2), (4, 65, 129, 512, 512, 3, 4), (4, 65, 129, 2048, 256, 3, 12), (8, 65, 129, 256, 256, 3, 2)]
for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    ops.activation_limbs(x)
    pad = dil * (k - 1) // 2
    out = torch.empty(B, H, W, Cout, device="cuda")
    ws = ops._conv_ws(x.device)
    for _ in range(2):
        ops.conv_forward(x, wp, k, 1, dil, pad, Cout, out=out, want_stats=4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_forward(x, wp, k, 1, dil, pad, Cout, out=out, want_stats=4)
    e1.record()
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[:256 * 4].reshape(256, 4).double().cpu()
    m = st.mean(0).tolist()
    print(f"Cin={Cin} Cout={Cout} k={k} d={dil} M={B*H*W}: event {e0.elapsed_time(e1)*1e3:.1f} us; mean ticks per workgroup: set-up {m[0]:.0f}  K loop {m[1]:.0f}  "
          f"epilogue + partial stores {m[2]:.0f}  whole {m[3]:.0f}   (max whole {st[:,3].max().item():.0f})")
