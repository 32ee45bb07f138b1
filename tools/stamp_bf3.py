"""Diagnostic: per-phase shader cycles of the bf16x3 forward kernel (in-kernel s_memtime stamps).

Builds the kernels with -DONDA_BF3_STAMP into tools/_stamp/libonda_hip.so (never the product
library), runs conv shapes one tile per workgroup (ONDA_CONV_SCHED=1) and prints the median cycles
per K-step each wave spends in: vmcnt wait | barrier 1 | split + store A | barrier 2 | issue of
the next loads / DMA | fragment reads + MFMA issue, plus the in-kernel shader clock
(d s_memtime / d wall_clock64 x 100 MHz).  Stamps cost ~10 % and MFMAs may slide across them, so
read the split between "vmcnt" and "reads+mfma" with that in mind.

    python tools/stamp_bf3.py build      (in the build container; `build clock` = only two stamps per
                                          tile: the in-kernel clock of the unperturbed loop)
    REPS=400 python tools/stamp_bf3.py   (on the GPU box; ZERO=1 feeds all-zero activations)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_stamp")
LIB = os.path.join(OUT, "libonda_hip.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for src in ["conv.hip", "conv_bf3.hip", "conv_h2.hip", "norm.hip", "pointwise.hip", "loss_proto.hip", "pipeline.hip"]:
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DONDA_BF3_CLOCK" if "clock" in sys.argv[2:] else "-DONDA_BF3_STAMP",
                               "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "onda_amd", "csrc", src), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    sys.exit(0)
os.environ["ONDA_LIB_PATH"] = LIB
os.environ["ONDA_CONV_SCHED"] = "1"
sys.path.insert(0, ROOT)
import torch
from onda_amd import ops
NAMES = ["vmcnt", "barrier1", "split+store", "barrier2", "issue loads", "reads+mfma"]
for (B, H, W, Cin, Cout, k, dil) in [(4, 65, 129, 512, 512, 3, 4), (4, 65, 129, 2048, 512, 1, 1), (4, 65, 129, 256, 1024, 1, 1)]:
    x = torch.randn(B, H, W, Cin, device="cuda")
    if os.environ.get("ZERO"):
        x.zero_()
    w = torch.randn(Cout, Cin, k, k, device="cuda") / (Cin * k * k) ** 0.5
    wp = ops.pack_weight_fwd(w)
    pad = dil * (k - 1) // 2
    reps = int(os.environ.get("REPS", 5))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.conv_forward(x, wp, k, 1, dil, pad, Cout)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    KT = k * k * Cin // 32
    w8 = ops._conv_ws(x.device)[:256 * 4 * 8].view(-1, 8).double().cpu()
    med = w8.median(dim=0).values[:6] / KT
    ok = w8[:, 7] > 0
    print(f"Cin={Cin} Cout={Cout} k={k}: {KT} K-steps; cycles per K-step per wave: " +
          "  ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, med.tolist())) + f"  | total {med.sum():.0f}"
          f"  | shader clock {(w8[ok, 6] / w8[ok, 7]).median().item() * 0.1:.2f} GHz"
          f"  | {us:.0f} us per launch = {2.0 * B * H * W * Cout * Cin * k * k / us / 1e6:.0f} TFLOP/s")
