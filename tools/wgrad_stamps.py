"""In-kernel timeline of conv_wgrad_l2_kernel (onda_debug_stamps): per workgroup s_memtime at the start, after the set-up
(live K-step list, prologue issue), after the K loop and after the slab store; prints mean cycles per phase and per K-step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops, _lib

SHAPES = [(4, 65, 129, 512, 512, 3, 4), (4, 65, 129, 2048, 256, 3, 12), (4, 65, 129, 256, 2048, 3, 24), (4, 65, 129, 1024, 256, 1, 1),
          (4, 65, 129, 2048, 512, 1, 1), (4, 65, 129, 256, 256, 3, 2), (4, 65, 129, 128, 128, 3, 1)]
ops.H2_PATH = "dma"
buf = torch.zeros(4096, 8, dtype=torch.int64, device="cuda")
lib = _lib.load()
for (B, H, W, Cin, Cout, k, dil) in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    dy = torch.randn(B, H, W, Cout, device="cuda")
    pad = dil * (k - 1) // 2
    ops.activation_limbs(x); ops.activation_limbs(dy)
    for _ in range(2):
        ops.conv_wgrad(x, dy, k, 1, dil, pad, Cout, Cin)
    torch.cuda.synchronize()
    buf.zero_()
    lib.onda_debug_stamps(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_wgrad(x, dy, k, 1, dil, pad, Cout, Cin)
    e1.record()
    torch.cuda.synchronize()
    lib.onda_debug_stamps(None)
    st = buf.cpu()
    st = st[st[:, 0] > 0]
    setup = (st[:, 1] - st[:, 0]).float()
    loop = (st[:, 2] - st[:, 1]).float()
    slab = (st[:, 3] - st[:, 2]).float()
    nlive = st[:, 5].float()
    span = (st[:, 3].max() - st[:, 0].min()).item()
    per = (loop / nlive.clamp_min(1))
    pk = st[:, 4]
    iss, rd, wt, cmp_ = ((pk >> 40) & 0xFFFFF).float(), ((pk >> 20) & 0xFFFFF).float(), (pk & 0xFFFFF).float(), st[:, 7].float()
    print(f"   one K-step of the first half (cycles): DMA issue {iss.mean():.0f} | fragment reads {rd.mean():.0f} | wait at the slot barrier "
          f"{wt.mean():.0f} | compute slot up to its end {cmp_.mean():.0f}")
    print(f"Cin={Cin} Cout={Cout} k={k} d={dil}: {st.shape[0]} workgroups (of the first 4096), event {e0.elapsed_time(e1)*1e3:.0f} us, span {span} ticks | "
          f"set-up {setup.mean():.0f} | K loop {loop.mean():.0f} ({nlive.mean():.0f} live steps, {per.mean():.0f} per step, min {per.min():.0f}) | slab store {slab.mean():.0f}")
