"""Phase stamps of conv_l2_kernel (ONDA_L2_DEBUG=5): cycles per workgroup in tile setup+prologue / K loop / epilogue."""
import os, sys
os.environ["ONDA_L2_DEBUG"] = "5"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops
SHAPES = [(4, 65, 129, 256, 2048, 3, 12), (4, 65, 129, 256, 1024, 1, 1), (4, 65, 129, 1024, 256, 1, 1), (4, 65, 129, 512, 2048, 1, 1)]
for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    pad = dil * (k - 1) // 2
    for _ in range(3):
        ops.conv_forward(x, wp, k, 1, dil, pad, Cout, want_stats=True)
    torch.cuda.synchronize()
    ws = ops._conv_ws(x.device)
    st = ws.view(torch.int64)[: 256 * 4].reshape(256, 4).cpu().double()
    M = B * H * W
    tiles = (M + 255) // 256 * ((Cout + 127) // 128)
    per_wg = tiles // 256
    KT = k * k * Cin // 32
    med = st.median(0).values
    print(f"Cin={Cin} Cout={Cout} k={k}: tiles/WG {per_wg} KT {KT} | per tile cycles: setup {med[0]/per_wg:8.0f} loop {med[1]/per_wg:8.0f} "
          f"(={med[1]/per_wg/KT:6.0f}/K-step; MFMA floor 1536) epilogue {med[2]/per_wg:8.0f} | total {med[3]:9.0f}")
