#!/usr/bin/env python3
"""Headline benchmark: adaptation-step throughput (target images / second) of the
hybrid-switch online adaptation step -- `hybrid_proDA.step([source], target)` +
`update_ema()` = 2x (forward+backward) + 2-3 no-grad forwards of DeepLabV2/ResNet-50 +
prototype pseudo-labelling + losses + SGD + teacher EMA -- on synthetic 512x1024 batches
of 4 images per GPU (BASELINE.json configs[2]; configs[3] when launched on N GPUs).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` counts all ranks' target images over the max-over-ranks
time of exactly K steps (inputs already resident in HBM).  Weak scaling: every rank adapts
on its own micro-batch of 4; gradients, prototype statistics and the switch scalars are
all-reduced over RCCL (onda_amd/dist.py).

`roofline` is measured live with HIP events on the launch stream in extra, instrumented
steps after the timed region; `cpu_baseline` times the CPU oracle (oracle/step.py, the
restatement of the reference pinned by the golden vectors) on the host cores, rank 0, N=1.
"""
import argparse
import glob
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# SURVEY 8d / BASELINE.md section 2: exact conv FLOPs per image at 512x1024
FWD_GFLOP_PER_IMG = 781.05
BWD_GFLOP_PER_IMG = 1559.64
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 (the dtype's dense matrix peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak: the pipe the split-precision kernels execute on


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step")
    ap.add_argument("--branch", choices=["dynamic", "static"], default="dynamic",
                    help="which side of the hybrid switch the synthetic state sits on (pinned via the head scale)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def build_adapter(args, device, tmp):
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.synthetic import fill_state_dict, synth_batch
    cfg, spec = hybrid_switch_cfg(args.width, args.height, device, tmp, batch_size=args.batch)
    torch.manual_seed(123)  # hybrid_switch.yml RANDOM_SEED
    model = get_model(cfg, 19)
    # confident static prior (> 0.9) -> static branch; diffuse one (< 0.83) -> dynamic branch
    fill_state_dict(model, 1, 40.0 if args.branch == "static" else 1.0)
    da = get_adapt_method(cfg)(model, cfg, spec)
    rank = int(os.environ.get("RANK", "0"))

    def dev_batch(seed):
        b = synth_batch(args.batch, args.height, args.width, seed=seed)
        return {k: v.to(device) for k, v in b.items()}

    src = [dev_batch(1000 + 10 * rank + i) for i in range(2)]
    trg = [dev_batch(2000 + 10 * rank + i) for i in range(2)]
    da.update_dynamic()
    switch_batch_statistics(da.model, False)
    da.calculate_prototypes(src, save=False)  # the reference's `append` path over 2 source batches
    switch_batch_statistics(da.model, True)
    da.optimizer.zero_grad()
    return da, src, trg


def one_step(da, src, trg, i, total):
    da.adjust_learning_rate(i, total)
    log = da.step([src[i % 2]], trg[i % 2])
    da.update_ema()
    return log


def measure_roofline(da, src, trg, args, steps_done, record=True):
    """Per-kernel-family time via events recorded around every conv launch on the launch
    stream (torch's current stream).  Reported for the dominant family: the 128x128-tile
    MFMA implicit-GEMM kernel that runs all forward and data-gradient convolutions.
    Every rank runs the two instrumented steps (they contain the step's collectives); only the
    recording rank keeps events."""
    from onda_amd import ops
    if record:
        ops.PROFILE = []
    one_step(da, src, trg, steps_done, steps_done + 2)
    one_step(da, src, trg, steps_done + 1, steps_done + 2)
    torch.cuda.synchronize()
    if not record:
        return None
    fam = {}
    for name, flops, e0, e1, _tag in ops.PROFILE:
        f = fam.setdefault(name, [0.0, 0.0, 0])
        f[0] += flops
        f[1] += e0.elapsed_time(e1) * 1e-3
        f[2] += 1
    ops.PROFILE = None
    dom = max(fam.items(), key=lambda kv: kv[1][1])
    name, (flops, secs, n) = dom
    achieved = flops / secs / 1e12
    detail = {k: {"launches": v[2], "ms_total": round(v[1] * 1e3, 3), "tflops": round(v[0] / v[1] / 1e12, 2)}
              for k, v in fam.items()}
    # The roofline of the kernel is the matrix pipe it executes on: `peak` = that pipe's dense peak divided by
    # the MFMA products the evaluation spends per fp32 product (f16x2: 3, bf16x3: 6, f32: the fp32 MFMA itself),
    # so `frac` = executed MFMA flops / pipe peak.  The comparison with the fp32 matrix peak (what a kernel
    # that did not split its operands could reach at most) is reported next to it.
    if "h2" in name:
        products, what = 3, "f16 MFMA (v_mfma_f32_16x16x32_f16), 3 limb products per fp32 product, fp32 accumulate"
    elif "bf3" in name:
        products, what = 6, "bf16 MFMA (v_mfma_f32_16x16x32_bf16), 6 limb products per fp32 product, fp32 accumulate"
    else:
        products, what = 0, "fp32 MFMA (v_mfma_f32_32x32x2_f32)"
    peak = BF16_MFMA_PEAK_TFLOPS / products if products else FP32_MFMA_PEAK_TFLOPS
    roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
            "launches_per_step": n // 2, "avg_launch_ms": round(secs / n * 1e3, 4),
            "avg_launch_gflop": round(flops / n / 1e9, 3), "families": detail,
            "pipe": {"what": what, "pipe_peak_tflops": BF16_MFMA_PEAK_TFLOPS if products else FP32_MFMA_PEAK_TFLOPS,
                     "executed_tflops": round((products or 1) * achieved, 1)},
            "vs_fp32_matrix_peak": {"peak": FP32_MFMA_PEAK_TFLOPS, "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)}}
    roof.update(committed_traffic(name))
    return roof


def committed_traffic(kernel_family):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3
    FETCH_SIZE / WRITE_SIZE in separate passes over this same bench.py; tools/pmc_summary.py
    applies the gfx950 corrections).  Counters cannot be read from inside the process, so the
    number is the last profiled one and says which file it came from."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json"))):
        try:
            data = json.load(open(path))["kernels"]
        except Exception:
            continue
        stem = kernel_family.split("<")[0]
        rows = [(k, v) for k, v in data.items() if stem + "<" in k]
        if rows:
            tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for _, v in rows)
            n = sum(v["launches"] for _, v in rows)
            best = {"traffic": round(tot / n), "traffic_unit": "bytes per launch (HBM + Infinity-Cache side of L2)",
                    "traffic_source": os.path.relpath(path, ROOT)}
    return best or {}


def cpu_baseline(args):
    """The CPU oracle on the host cores: one full step (+update_ema) on a bounded sample."""
    from onda_amd.synthetic import synth_batch, synth_tensor
    from oracle import model as omodel
    from oracle.step import OracleAdapter
    # torch's CPU convolutions stop scaling (and then collapse) far below the 256 hardware threads
    # of the GPU node's host; 32 threads is about the best it does, and it is what is reported
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    hs = 40.0 if args.branch == "static" else 1.0
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, hs).to(dt) for k, shape, dt in omodel.state_spec()}
    b, h, w = 1, args.height, args.width
    src, trg = synth_batch(b, h, w, seed=1000), synth_batch(b, h, w, seed=2000)
    from onda_amd.synthetic import synth_prototypes
    ad = OracleAdapter(sd, synth_prototypes())
    ad.refresh_dynamic()
    masks = tuple(omodel.draw_drop_mask(b) for _ in range(3))
    t0 = time.perf_counter()
    ad.step(src, trg, masks)
    ad.update_ema()
    dt = time.perf_counter() - t0
    return {"value": round(b / dt, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 hybrid step (+update_ema) of the CPU oracle, B={b} at {w}x{h}, "
                      f"branch={'dynamic' if ad.switch.current else 'static'}, {dt:.1f} s, torch CPU fp32, {cores} threads"}


def conv_accuracy_probe(device):
    """Evidence for the line's `dtype`: the conv path's error against fp64 on a sample (3x3 dilated conv,
    forward + both gradients), next to the error of a plain fp32 conv (torch CPU) on the same data."""
    import torch.nn.functional as F
    from onda_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(2, 128, 17, 23, generator=g)) * torch.exp(torch.randn(2, 128, 17, 23, generator=g))
    w = torch.randn(128, 128, 3, 3, generator=g) * 0.03
    gy = torch.randn(2, 128, 17, 23, generator=g)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(x64, w64, None, 1, 2, 2).backward(gy.double())
    x32, w32 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(x32, w32, None, 1, 2, 2).backward(gy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(device).requires_grad_(True)
    wd = w.to(device).requires_grad_(True)
    y, _ = ops.Conv2dFn.apply(xd, wd, None, ops._PackCache(), 1, 2, 2, False, None)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(device))

    def rel(a, ref):
        return float((a.double().cpu() - ref).norm() / ref.norm())

    return {"dgrad_rel_l2_vs_fp64": rel(xd.grad.permute(0, 3, 1, 2), x64.grad), "wgrad_rel_l2_vs_fp64": rel(wd.grad, w64.grad),
            "torch_cpu_fp32_dgrad_rel_l2_vs_fp64": rel(x32.grad, x64.grad),
            "torch_cpu_fp32_wgrad_rel_l2_vs_fp64": rel(w32.grad, w64.grad)}


def conv_mode_note():
    from onda_amd import ops
    if ops.CONV_MODE == "f16x2":
        return ("f16x2: fp32 operands scaled by a per-tensor power of two and split into 2 f16 limbs, 3 limb products on "
                "the f16 MFMA pipe, fp32 accumulation (3e-7 relative L2 against fp64, the accuracy of an fp32 FMA chain; "
                "same parity tests as the exact-fp32 MFMA kernels)")
    if ops.CONV_MODE == "bf16x3":
        return ("bf16x3: fp32 operands split exactly into 3 bf16 limbs, 6 limb products on the bf16 MFMA pipe, "
                "fp32 accumulation (2e-7 relative to the exact-fp32 MFMA kernels; same parity tests)")
    return "f32: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain)"


def main():
    args = parse()
    from onda_amd import dist as odist
    rank, world, local = odist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    with tempfile.TemporaryDirectory() as tmp:
        da, src, trg = build_adapter(args, device, tmp)
        total = args.warmup + args.steps
        for i in range(args.warmup):
            one_step(da, src, trg, i, total)
        odist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            log = one_step(da, src, trg, args.warmup + i, total)
        torch.cuda.synchronize()
        odist.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        odist.all_reduce_max(tmax)
        dt = tmax.item()
        branch = "dynamic" if da.model_select.current == 1 else "static"
        roof = None
        if not args.no_roofline:
            roof = measure_roofline(da, src, trg, args, total, record=(rank == 0))
        loss = float(log["Total target loss"].detach())
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    if rank == 0:
        n_fwd = 2 + (1 if branch == "dynamic" else 0)
        tflop_step = args.batch * (2 * (FWD_GFLOP_PER_IMG + BWD_GFLOP_PER_IMG) + n_fwd * FWD_GFLOP_PER_IMG) / 1e3
        if (args.height, args.width) != (512, 1024):
            tflop_step = None
        line = {
            "metric": f"adaptation-step images/sec (fwd+bwd+proto) {args.height}x{args.width} bs={args.batch}",
            "value": round(world * args.batch * args.steps / dt, 4), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"hybrid_switch adaptation step (step + update_ema), {args.width}x{args.height}, "
                                   f"bs={args.batch} per GPU, {branch} branch, DeepLabV2-ResNet50 ProDA head, "
                                   f"random-init weights", "global_batch": world * args.batch,
                       "parallelism": f"dp{world}", "branch": branch, "conv_mode": conv_mode_note(),
                       "conv_tflop_per_step_per_gpu": tflop_step, "final_loss": round(loss, 5),
                       "conv_accuracy": conv_accuracy_probe(device)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if tflop_step:
            line["config"]["step_conv_tflops_per_gpu"] = round(tflop_step / (dt / args.steps), 2)
        print(json.dumps(line), flush=True)
    if odist.is_on():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
