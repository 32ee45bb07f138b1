#!/usr/bin/env python3
"""Headline benchmark: adaptation-step throughput (target images / second) of the hybrid-switch online adaptation
step -- `hybrid_proDA.step([source], target)` + `update_ema()` = 2x (forward+backward) + 2-3 no-grad forwards of
DeepLabV2/ResNet-50 + prototype pseudo-labelling + losses + SGD + teacher EMA -- on synthetic 512x1024 batches of 4
images per GPU (BASELINE.json configs[2]; configs[3] when launched on N GPUs).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      # without a launcher: starts the N ranks itself (fresh child processes), or fails

Rank 0 prints ONE JSON line.  `value` counts all ranks' target images over the max-over-ranks time of exactly K steps
(inputs already resident in HBM).  Default = weak scaling: every rank adapts on its own micro-batch of 4; the
exchange is onda_amd/dist.py (two collectives per step, the gradient buckets overlapped with the last backward).
`--global-batch 32` = strong scaling: a fixed global batch processed by N GPUs, each running
32 / (4 N) micro-batches per optimizer step with the arithmetic of that many more ranks (`step_sharded`).
Other BASELINE configs: `--config 1` (forward-only evaluation of 8 frames), `--config 2` (supervised fwd+bwd+CE+SGD
step), `--config 5` (the adaptation step at 1024x2048).

The line carries its own context: `dtype` names the arithmetic the convolutions run in; `config.exact_f32` re-times 3
steps with the exact fp32-MFMA kernels; `config.eager_rocm` times the same step on PyTorch-ROCm eager (MIOpen fp32) in
this process -- live by default when MIOpen's tuned find-db travels with the tree (tools/miopen_db: seconds), with
`--eager` otherwise (MIOpen's kernel search: ~10 minutes), and `vs_baseline` = value / that live eager rate;
`config.other_configs` holds short driver-timed runs of BASELINE configs 1, 2, 5 and of the static branch; `roofline`
is measured live with HIP events on the launch stream in extra, instrumented steps after the timed region;
`cpu_baseline` times the CPU oracle (oracle/step.py, the restatement of the reference pinned by the golden vectors) on
the host cores at the GPU line's batch size, rank 0, N=1; `config.library` ties the binary to the sources (hash
compiled into libonda_hip.so vs hash of the sources on disk).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# SURVEY 8d / BASELINE.md section 2: exact conv FLOPs per image
CONV_GFLOP = {(512, 1024): (781.05, 1559.64), (1024, 2048): (3088.60, 6167.33)}
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 (the dtype's dense matrix peak)
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak: the pipe the split-precision kernels execute on


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 5], help="BASELINE.json configs[N-1]")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per micro-batch")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: fixed global batch per optimizer step")
    ap.add_argument("--eval-batch", type=int, default=8, choices=[1, 2, 4, 8], help="config 1: frames per forward pass")
    ap.add_argument("--branch", choices=["dynamic", "static"], default="dynamic",
                    help="which side of the hybrid switch the synthetic state sits on (pinned via the head scale)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true",
                    help="time the step on PyTorch-ROCm eager in this process even without a tuned MIOpen find-db in the tree "
                         "(MIOpen's kernel search then takes ~10 minutes)")
    ap.add_argument("--no-eager", action="store_true")
    ap.add_argument("--eager-only", action="store_true", help=argparse.SUPPRESS)  # the child process of the time-boxed eager leg
    ap.add_argument("--no-exact-f32", action="store_true")
    ap.add_argument("--h2d", action="store_true",
                    help="the batches start in pinned HOST memory and are uploaded every step on a copy stream (the drop-in's "
                         "real input path); the default line reports this as a second number, config.h2d")
    ap.add_argument("--no-h2d-leg", action="store_true")
    ap.add_argument("--no-exchange-probe", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of BASELINE configs 1, 2, 5 and of the static branch (config.other_configs)")
    args = ap.parse_args()
    if args.height is None:
        args.height, args.width = (1024, 2048) if args.config == 5 else (512, 1024)
    return args


# ------------------------------------------------------------------------------------------------- workloads
def build_adapter(args, device, tmp, shards=1):
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

    n = 2 * shards
    src = [dev_batch(1000 + 100 * rank + i) for i in range(n)]
    trg = [dev_batch(2000 + 100 * rank + i) for i in range(n)]
    da.update_dynamic()
    switch_batch_statistics(da.model, False)
    da.calculate_prototypes(src[:2], save=False)  # the reference's `append` path over 2 source batches
    switch_batch_statistics(da.model, True)
    da.optimizer.zero_grad()
    return da, src, trg


def fresh(batch):
    """The batch as a loader would deliver it: NEW tensor objects every step (same device storage -- inputs stay resident
    in HBM).  Anything the operator layer remembers on an input tensor (ops.stem_patches caches the stem's patch matrix on
    the image tensor, so that teacher / static / student share it WITHIN a step) therefore never survives into the next
    step: every timed step pays its absmax + patch passes (round-2 verdict: 4 patch launches in 10 steps)."""
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in batch.items()}


class HostFeeder:
    """The input path of a real run: batches live in pinned host memory (a DataLoader's pin_memory) and are uploaded once per
    step -- ONE upload of each batch (the reference uploads the target batch twice: prototypes.py:283,
    prototypes_hybrid_switch.py:48) -- on a copy stream, step i + 1's while step i computes."""

    def __init__(self, src, trg, device):
        self.device = device
        self.host = [tuple({k: v.cpu().pin_memory() for k, v in b.items()} for b in (s_, t_)) for s_, t_ in zip(src, trg)]
        self.stream = torch.cuda.Stream(device=device)
        self.next = None
        self.bytes_per_step = sum(v.numel() * v.element_size() for b in self.host[0] for v in b.values())

    def _upload(self, i):
        pair = self.host[i % len(self.host)]
        with torch.cuda.stream(self.stream):
            dev = tuple({k: v.to(self.device, non_blocking=True) for k, v in b.items()} for b in pair)
            done = torch.cuda.Event()
            done.record(self.stream)
        return dev, done

    def get(self, i):
        if self.next is None or self.next[0] != i:
            self.next = (i,) + self._upload(i)
        _, dev, done = self.next
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        for b in dev:
            for v in b.values():
                v.record_stream(cur)
        self.next = (i + 1,) + self._upload(i + 1)
        return dev


def one_step(da, src, trg, i, total, shards=1, feeder=None):
    da.adjust_learning_rate(i, total)
    if feeder is not None:
        s_, t_ = feeder.get(i)
        log = da.step([s_], t_)
    elif shards == 1:
        log = da.step([fresh(src[i % 2])], fresh(trg[i % 2]))
    else:
        base = (i % 2) * shards
        log = da.step_sharded([([fresh(src[base + j])], fresh(trg[base + j])) for j in range(shards)])
    da.update_ema()
    return log


def timed_loop(fn, warmup, steps, device):
    from onda_amd import dist as odist
    for i in range(warmup):
        fn(i)
    odist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        out = fn(warmup + i)
    torch.cuda.synchronize()
    odist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    odist.all_reduce_max(tmax)
    return tmax.item(), out


# ------------------------------------------------------------------------------------------------- roofline
def measure_roofline(step_fn, steps_done, record=True):
    """Per-kernel-family time via events recorded around every conv launch on the launch stream (torch's current
    stream).  Reported for the dominant family.  Every rank runs the two instrumented steps (they contain the step's
    collectives); only the recording rank keeps events."""
    from onda_amd import ops
    from onda_amd.framework.domain_adaptation.methods import prototypes as pmod
    # A kernel's roofline fraction is a property of the kernel: the instrumented steps run with the no-grad passes back on
    # the main stream (ONDA_SIDE_STREAMS=0), so that an event pair brackets ONE kernel that has the GPU to itself -- beside
    # other streams' launches the pair would also time the wait for the CUs they hold (and rocprof's per-kernel durations
    # stretch the same way).  The matching rocprofv3 summary is profiles/*_bench_kernel_stats_one_stream.csv.
    side, pmod.SIDE_STREAMS = pmod.SIDE_STREAMS, False
    try:
        step_fn(steps_done)  # (not recorded: the first step after the switch re-allocates workspaces on the main stream)
        if record:
            ops.PROFILE = []
        step_fn(steps_done + 1)
        step_fn(steps_done + 2)
        torch.cuda.synchronize()
    finally:
        pmod.SIDE_STREAMS = side
    if not record:
        return None
    fam = {}
    for name, flops, e0, e1, tag, issued in ops.profile_entries(ops.PROFILE):
        f = fam.setdefault(name, [0.0, 0.0, 0, 0.0, 0.0])
        f[0] += flops
        f[1] += e0.elapsed_time(e1) * 1e-3
        f[2] += 1
        f[3] += issued
        f[4] += algorithmic_bytes(tag)
    ops.PROFILE = None
    name, (flops, secs, n, issued, abytes) = max(fam.items(), key=lambda kv: kv[1][1])
    achieved = flops / secs / 1e12
    detail = {k: {"launches": v[2], "ms_total": round(v[1] * 1e3, 3), "tflops": round(v[0] / v[1] / 1e12, 2),
                  "issued_share": round(v[3] / v[0], 4)} for k, v in fam.items()}
    fwd = [v for k, v in fam.items() if "wgrad" not in k]
    wg = [v for k, v in fam.items() if "wgrad" in k]
    # The roofline of a kernel is the matrix pipe it executes on: `peak` = that pipe's dense peak divided by the MFMA
    # products the evaluation spends per fp32 product (f16x2: 3, f32: the fp32 MFMA itself).  `achieved` / `frac` are in
    # ALGORITHMIC flops (2 M Cout taps Cin per launch); the kernels skip K-steps that only multiply padding
    # (dead taps of the dilated branches), so the matrix pipe itself executes `pipe.executed_tflops` =
    # products x ISSUED flops / time (< products x achieved).  The fp32 matrix peak (what an unsplit fp32 kernel could
    # reach at most) is reported next to it.
    if "l2" in name or "h2" in name:
        products, what = 3, "f16 MFMA (v_mfma_f32_16x16x32_f16), 3 limb products per fp32 product, fp32 accumulate"
    else:
        products, what = 0, "fp32 MFMA (v_mfma_f32_32x32x2_f32)"
    peak = F16_MFMA_PEAK_TFLOPS / products if products else FP32_MFMA_PEAK_TFLOPS
    pipe_peak = F16_MFMA_PEAK_TFLOPS if products else FP32_MFMA_PEAK_TFLOPS
    executed = (products or 1) * issued / secs / 1e12
    roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "measured": "HIP events around every launch of two extra steps with ONDA_SIDE_STREAMS=0 (each kernel alone on the GPU)",
            "schedule": "one_stream (a per-kernel property; ms_per_step / value come from the default multi-stream schedule)",
            "frac": round(achieved / peak, 4), "traffic": None, "launches_per_step": n // 2,
            "avg_launch_ms": round(secs / n * 1e3, 4), "avg_launch_gflop": round(flops / n / 1e9, 3),
            "avg_launch_algorithmic_bytes": round(abytes / n), "families": detail,
            "all_fwd_dgrad_tflops": round(sum(v[0] for v in fwd) / max(sum(v[1] for v in fwd), 1e-12) / 1e12, 2) if fwd else None,
            "all_wgrad_tflops": round(sum(v[0] for v in wg) / max(sum(v[1] for v in wg), 1e-12) / 1e12, 2) if wg else None,
            "pipe": {"what": what, "pipe_peak_tflops": pipe_peak, "issued_share_of_algorithmic_flops": round(issued / flops, 4),
                     "executed_tflops": round(executed, 1), "executed_frac_of_pipe_peak": round(executed / pipe_peak, 4)},
            "vs_fp32_matrix_peak": {"peak": FP32_MFMA_PEAK_TFLOPS, "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)}}
    roof.update(committed_traffic(name))
    if roof.get("traffic"):
        # bytes that crossed the L2's memory side per launch / bytes the launch has to touch once (operands + output):
        # > 1 = re-reads (the nine taps of a 3x3 convolution re-streaming their input rows through a 4 MB L2)
        roof["traffic_over_algorithmic_bytes"] = round(roof["traffic"] / max(roof["avg_launch_algorithmic_bytes"], 1), 3)
    return roof


def algorithmic_bytes(tag):
    """Bytes a conv launch has to touch once: its operands (limb planes: 4 bytes per element, like fp32) and its output.
    tag = (kind, M, Cout, Cin, k, stride, dil[, splitk]) as ops records it; M = GEMM rows of the launch."""
    if not tag:
        return 0.0
    kind, M, co, ci, k = tag[0], tag[1], tag[2], tag[3], tag[4]
    if kind == "wgrad":
        return 4.0 * (M * ci + M * co) + 4.0 * tag[7] * co * k * k * ci
    return 4.0 * (M * ci + co * k * k * ci + M * co)


def committed_traffic(kernel_family):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3 FETCH_SIZE / WRITE_SIZE in
    separate passes over this same bench.py; tools/pmc_summary.py applies the gfx950 corrections).  Counters cannot be
    read from inside the process, so the number is the last profiled one; the file says which source hash it was
    taken on (`library_src`), to be compared with `config.library`."""
    best = None
    # bench names ("conv_l2_kernel<4,2>", "conv_l2x_kernel<4,2>", ...) -> the prefix of the demangled name in the profile
    stem = kernel_family.replace(",", ", ").rstrip(">") if kernel_family.startswith("conv_l2") else kernel_family.split("<")[0] + "<"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json"))):
        try:
            blob = json.load(open(path))
            data = blob["kernels"]
        except Exception:
            continue
        rows = [(k, v) for k, v in data.items() if stem in k]
        if rows:
            tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for _, v in rows)
            n = sum(v["launches"] for _, v in rows)
            best = {"traffic": round(tot / n), "traffic_unit": "bytes per launch (HBM + Infinity-Cache side of L2)",
                    "traffic_source": os.path.relpath(path, ROOT), "traffic_library_src": blob.get("library_src"),
                    "traffic_workload": blob.get("workload", "tools/one_pass.py (one bs-4 forward+backward pass; NOT the bench step)"),
                    "traffic_avg_launch_us_profiled": round(sum(v["avg_launch_us_profiled"] * v["launches"] for _, v in rows) / n, 1)}
    return best or {}


# ------------------------------------------------------------------------------------------------- baselines
def _oracle_adapter(args, device="cpu", batch=1):
    from onda_amd.synthetic import synth_batch, synth_prototypes, synth_tensor
    from oracle import model as omodel
    from oracle.step import OracleAdapter
    hs = 40.0 if args.branch == "static" else 1.0
    sd = {k: synth_tensor(k, torch.empty(shape, dtype=dt), 1, hs).to(dt).to(device) for k, shape, dt in omodel.state_spec()}
    ad = OracleAdapter(sd, tuple(t.to(device) for t in synth_prototypes()))
    ad.refresh_dynamic()
    src = {k: v.to(device) for k, v in synth_batch(batch, args.height, args.width, seed=1000).items()}
    trg = {k: v.to(device) for k, v in synth_batch(batch, args.height, args.width, seed=2000).items()}
    return ad, src, trg, omodel


def cpu_baseline(args):
    """The CPU oracle on the host cores: full steps (+update_ema) on a bounded sample: one warm-up, one timed."""
    # torch's CPU convolutions stop scaling (and then collapse) far below the 256 hardware threads of the GPU node's
    # host; 32 threads is about the best it does, and it is what is reported
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    b = args.batch  # the GPU line's unit: same batch (same BatchNorm batch), same step
    ad, src, trg, omodel = _oracle_adapter(args, "cpu", b)
    times = []
    for _ in range(2):
        masks = tuple(omodel.draw_drop_mask(b) for _ in range(3))
        t0 = time.perf_counter()
        ad.step(src, trg, masks)
        ad.update_ema()
        times.append(time.perf_counter() - t0)
    dt = times[-1]
    return {"value": round(b / dt, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"hybrid step (+update_ema) of the CPU oracle on {b} image(s) per step at {args.width}x{args.height} "
                      f"(the GPU line's batch), branch={'dynamic' if ad.switch.current else 'static'}; 1 warm-up "
                      f"({times[0]:.1f} s) + 1 timed step ({dt:.1f} s), torch CPU fp32, {cores} threads"}


MIOPEN_DB = os.path.join(ROOT, "tools", "miopen_db")  # MIOpen's user find-db + kernel cache of THIS workload, in-tree


def miopen_db_ready():
    """Has MIOpen's kernel search for this workload's convolutions been done before (`python bench.py --eager` on an MI355X
    fills tools/miopen_db) and do its results travel with the tree?  Then the live eager leg costs seconds instead of ten minutes."""
    return bool(glob.glob(os.path.join(MIOPEN_DB, "*.ufdb.txt")) or glob.glob(os.path.join(MIOPEN_DB, "*.udb.txt")))


def point_miopen_at_tree():
    """MIOpen reads these when its first handle is created (the first eager convolution; the HIP path never calls
    MIOpen).  The directory must be writable: a miss makes MIOpen search and append."""
    os.makedirs(MIOPEN_DB, exist_ok=True)
    os.environ.setdefault("MIOPEN_USER_DB_PATH", MIOPEN_DB)  # (honoured by a child only if it does not set its own: eager_rocm_boxed)
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(MIOPEN_DB, "cache"))


def eager_rocm(args, device):
    """The same step executed by PyTorch-ROCm eager kernels (MIOpen fp32 convolutions, torch's own BN / losses / SGD
    for-loop) on this GPU: the oracle restatement moved to the device.  MIOpen is allowed to pick its fastest kernels
    (cudnn.benchmark = True: the strongest eager baseline).  The reference itself sets benchmark = False and
    deterministic = True (train_ouda.py:28-30); with those flags MIOpen falls back to much slower kernels -- measured
    once on this workload: 11.3 s per step (profiles/r02_c_bench_line.json) -- so that setting is not the comparison."""
    import oracle.prototypes as op
    torch.backends.cudnn.benchmark = True
    torch.backends.cudnn.deterministic = False
    ad, src, trg, omodel = _oracle_adapter(args, device, args.batch)
    ones = torch.ones
    op.torch.ones = lambda *a, **k: ones(*a, **{**k, "device": device})  # the oracle's distance scratch lives on the CPU
    t_begin = time.perf_counter()
    try:
        def one():
            masks = tuple(omodel.draw_drop_mask(args.batch, device=device) for _ in range(3))
            ad.step(src, trg, masks)
            ad.update_ema()
        for _ in range(2):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        # the same with both batches uploaded from pinned host memory every step (as config.h2d of the HIP leg)
        hs = {k: v.cpu().pin_memory() for k, v in src.items()}
        ht = {k: v.cpu().pin_memory() for k, v in trg.items()}

        def one_h2d():
            s_ = {k: v.to(device, non_blocking=True) for k, v in hs.items()}
            t_ = {k: v.to(device, non_blocking=True) for k, v in ht.items()}
            masks = tuple(omodel.draw_drop_mask(args.batch, device=device) for _ in range(3))
            ad.step(s_, t_, masks)
            ad.update_ema()
        one_h2d()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            one_h2d()
        torch.cuda.synchronize()
        dt_h2d = (time.perf_counter() - t0) / n
    finally:
        op.torch.ones = ones
    return {"ms_per_step": round(dt * 1e3, 2), "images_per_s": round(args.batch / dt, 3),
            "h2d_ms_per_step": round(dt_h2d * 1e3, 2),
            "branch": "dynamic" if ad.switch.current else "static", "leg_seconds": round(time.perf_counter() - t_begin, 1),
            "what": "oracle step on PyTorch-ROCm eager (MIOpen fp32, cudnn.benchmark=True; 2 warm-up + 3 timed steps)"}


def eager_rocm_boxed(args):
    """The eager leg in a CHILD process with a time budget (ONDA_EAGER_BUDGET_S, default 420 s).  With the tuned find-db of
    tools/miopen_db it takes about a minute; if the db does not match the box's MIOpen after all -- another version, another
    device string -- MIOpen would search for ten minutes inside the driver's run: the child is then stopped and the committed
    measurement quoted instead.  (A fresh child that initialises the GPU itself; the parent has released its cached memory.)"""
    budget = float(os.environ.get("ONDA_EAGER_BUDGET_S", "420"))
    cmd = [sys.executable, os.path.abspath(__file__), "--eager-only", "--batch", str(args.batch), "--height", str(args.height),
           "--width", str(args.width), "--branch", args.branch]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # MIOpen appends to its user db and kernel cache whenever it meets something new: the child works on a COPY of the tracked
    # directory (1.7 MB), so that a measurement run never dirties the tree (`python bench.py --eager` is the run that fills it)
    import shutil
    scratch = tempfile.mkdtemp(prefix="onda_miopen_")
    shutil.copytree(MIOPEN_DB, os.path.join(scratch, "db"))
    env["MIOPEN_USER_DB_PATH"] = os.path.join(scratch, "db")
    env["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(scratch, "db", "cache")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        out, _ = child.communicate(timeout=budget)
    except subprocess.TimeoutExpired:
        child.kill()  # (this exact child, by handle)
        child.communicate()
        return None, f"the eager leg did not finish within {budget:.0f} s (MIOpen searching: tools/miopen_db does not match this box)"
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    if child.returncode != 0 or not lines:
        return None, f"the eager leg's process ended with code {child.returncode}"
    return json.loads(lines[-1]), None


def exchange_probe(args, plain_ms):
    """What the multi-GPU exchange costs a step, as far as ONE GPU can say: the same bench in a child process with
    ONDA_DIST_FORCE=1 -- one rank that runs the whole protocol of onda_amd/dist.py over RCCL (flat gradient views, bucketed
    all-reduce hooked into the last backward pass, the tail collective with prototype statistics / monitor scalars / running
    statistics, the division by the world size inside SGD) -- against this process's own step time.  Bytes moved over xGMI by
    a real ring are not in it; the launch, hook and bucket-copy overheads are."""
    budget = float(os.environ.get("ONDA_EXCHANGE_BUDGET_S", "180"))
    n = max(5, args.steps // 2)
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(n), "--warmup", "3", "--batch", str(args.batch), "--height",
           str(args.height), "--width", str(args.width), "--branch", args.branch, "--no-cpu-baseline", "--no-eager", "--no-other-configs",
           "--no-exact-f32", "--no-roofline", "--no-h2d-leg", "--no-exchange-probe"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"ONDA_DIST_FORCE": "1", "MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        out, _ = child.communicate(timeout=budget)
    except subprocess.TimeoutExpired:
        child.kill()  # (this exact child, by handle)
        child.communicate()
        return {"error": f"the forced-exchange run did not finish within {budget:.0f} s"}
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    if child.returncode != 0 or not lines:
        return {"error": f"the forced-exchange run ended with code {child.returncode}"}
    forced = json.loads(lines[-1])["ms_per_step"]
    return {"exchange_ms_per_step": round(forced - plain_ms, 3), "forced_ms_per_step": forced, "plain_ms_per_step": round(plain_ms, 3),
            "steps": n, "what": "ONDA_DIST_FORCE=1, one RCCL rank, child process on the same GPU, against this run's own step time"}


def conv_accuracy_probe(device):
    """Evidence for the line's `dtype`: the conv path's error against fp64 on a sample (3x3 dilated conv, forward + both
    gradients), next to the error of a plain fp32 conv (torch CPU) on the same data."""
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


def arithmetic():
    """(dtype label, note) of the convolution arithmetic in force."""
    from onda_amd import ops
    if ops.CONV_MODE == "f16x2":
        path = "both operands pre-split into limb rows in HBM, LDS-DMA only (conv_l2.hip)"
        return ("f32 (f16x2 split emulation)",
                "f16x2: fp32 operands scaled by a per-tensor power of two and split into 2 f16 limbs (22 significant bits), 3 limb "
                "products on the f16 MFMA pipe, fp32 accumulation; 1-3e-7 relative L2 against fp64 = the accuracy of an fp32 FMA "
                "chain, full accuracy for elements down to 2^-28 of a tensor's maximum; " + path)
    return ("f32", "f32: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain)")


def library_identity():
    from onda_amd import _lib, build
    built, src = build.check_fresh()
    return {"onda_version": _lib.query("onda_version").decode(), "library_src": built, "sources_on_disk": src, "fresh": built == src}


# ------------------------------------------------------------------------------------------------- configs
def run_adaptation(args, device, rank, world):
    from onda_amd import ops
    shards = 1
    if args.global_batch:
        per_step = args.batch * world
        if args.global_batch % per_step:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of {args.batch} x {world} ranks")
        shards = args.global_batch // per_step
    with tempfile.TemporaryDirectory() as tmp:
        da, src, trg = build_adapter(args, device, tmp, shards)
        total = args.warmup + args.steps
        feeder = HostFeeder(src, trg, device) if (args.h2d and shards == 1) else None
        step = lambda i: one_step(da, src, trg, i, total + 8, shards, feeder)  # noqa: E731
        dt, log = timed_loop(step, args.warmup, args.steps, device)
        branch = "dynamic" if da.model_select.current == 1 else "static"
        h2d = None
        if not args.h2d and not args.no_h2d_leg and shards == 1 and rank == 0 and world == 1 and not args.no_roofline:
            # the same step fed from pinned host memory, a second number beside the resident-input `value` (never `value` itself)
            fd = HostFeeder(src, trg, device)
            hstep = lambda i: one_step(da, src, trg, i, total + 8, 1, fd)  # noqa: E731
            n = max(3, args.steps // 2)
            hdt, _ = timed_loop(hstep, 1, n, device)
            h2d = {"ms_per_step": round(hdt / n * 1e3, 3), "images_per_s": round(args.batch * n / hdt, 3), "steps": n,
                   "uploaded_bytes_per_step": fd.bytes_per_step,
                   "what": "the step with both batches uploaded from pinned host memory every step (one upload each, on a copy stream, "
                           "step i + 1's under step i); `value` keeps the inputs resident in HBM"}
            del fd
        roof = None if args.no_roofline else measure_roofline(step, total, record=(rank == 0))
        loss = float(log["Total target loss"].detach())
        exact = None
        if not args.no_exact_f32 and rank == 0 and world == 1 and ops.CONV_MODE != "f32":
            old, ops.CONV_MODE = ops.CONV_MODE, "f32"
            try:
                step(total + 2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(3):
                    step(total + 3 + i)
                torch.cuda.synchronize()
                e = (time.perf_counter() - t0) / 3
                exact = {"ms_per_step": round(e * 1e3, 2), "images_per_s": round(args.batch * shards / e, 3),
                         "what": "same step, exact fp32 MFMA kernels (ONDA_CONV_MODE=f32), 3 steps"}
            finally:
                ops.CONV_MODE = old
    images = world * args.batch * shards
    gf = CONV_GFLOP.get((args.height, args.width))
    tflop_step = None
    if gf:
        n_fwd = 2 + (1 if branch == "dynamic" else 0)
        tflop_step = args.batch * shards * (2 * (gf[0] + gf[1]) + n_fwd * gf[0]) / 1e3
    cfg = {"workload": f"hybrid_switch adaptation step (step + update_ema), {args.width}x{args.height}, bs={args.batch} per "
                       f"micro-batch, {shards} micro-batch(es) per GPU and optimizer step, {branch} branch, DeepLabV2-ResNet50 "
                       f"ProDA head, random-init weights",
           "baseline_config": 5 if (args.height, args.width) == (1024, 2048) else (4 if world > 1 else 3),
           "global_batch": images, "parallelism": f"dp{world}", "branch": branch, "micro_batches_per_gpu": shards,
           "conv_tflop_per_step_per_gpu": tflop_step, "final_loss": round(loss, 5), "exact_f32": exact,
           "inputs": "uploaded from pinned host memory every step (--h2d)" if args.h2d else "resident in HBM", "h2d": h2d}
    if tflop_step:
        cfg["step_conv_tflops_per_gpu"] = round(tflop_step / (dt / args.steps), 2)
    return {"metric": f"adaptation-step images/sec (fwd+bwd+proto) {args.height}x{args.width} bs={args.batch}",
            "value": round(images * args.steps / dt, 4), "unit": "images/s", "dt": dt,
            "scaling": "strong" if args.global_batch else "weak", "config": cfg, "roofline": roof}


def run_segmentation(args, device, rank, world):
    """BASELINE config 2: `segmentation.train`'s step (segmentation.py:66-88) -- train-mode forward, bilinear upsample to
    the label resolution, cross-entropy, backward, SGD -- on 4 synthetic images per GPU."""
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods.segmentation import SegmentationTrainer
    from onda_amd.framework.handlers import get_model
    from onda_amd.synthetic import fill_state_dict, synth_batch
    cfg, spec = hybrid_switch_cfg(args.width, args.height, device, "NONE", batch_size=args.batch)
    spec.LEARNING_RATE, spec.POWER, spec.WEIGHT_DECAY = 2.5e-4, 0.9, 5e-4
    torch.manual_seed(123)
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, 3.0)
    model.train()
    tr = SegmentationTrainer(model, cfg, spec)
    batches = [{k: v.to(device) for k, v in synth_batch(args.batch, args.height, args.width, seed=3000 + 10 * rank + i).items()}
               for i in range(2)]
    step = lambda i: tr.step(batches[i % 2], 1000)  # noqa: E731
    dt, loss = timed_loop(step, args.warmup, args.steps, device)
    roof = None if args.no_roofline else measure_roofline(step, args.warmup + args.steps, record=(rank == 0))
    gf = CONV_GFLOP.get((args.height, args.width))
    tf = args.batch * (gf[0] + gf[1]) / 1e3 if gf else None
    cfg_out = {"workload": f"segmentation.train step (fwd -> bilinear upsample -> CE -> bwd -> SGD), {args.width}x{args.height}, "
                           f"bs={args.batch} per GPU, DeepLabV2-ResNet50 ProDA head, random-init weights",
               "baseline_config": 2, "global_batch": world * args.batch, "parallelism": f"dp{world}",
               "conv_tflop_per_step_per_gpu": tf, "final_loss": round(float(loss), 5)}
    if tf:
        cfg_out["step_conv_tflops_per_gpu"] = round(tf / (dt / args.steps), 2)
    return {"metric": f"supervised-step images/sec (fwd+bwd+CE+SGD) {args.height}x{args.width} bs={args.batch}",
            "value": round(world * args.batch * args.steps / dt, 4), "unit": "images/s", "dt": dt, "scaling": "weak",
            "config": cfg_out, "roofline": roof}


def run_forward_only(args, device, rank, world):
    """BASELINE config 1: the evaluation path (adaptation_model.py:127-166) -- eval-mode forward of 8 frames, fused
    upsample -> argmax -> confusion matrix; one "step" = the 8 frames."""
    from onda_amd import ops
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.handlers import get_model
    from onda_amd.synthetic import fill_state_dict, synth_batch
    cfg, _ = hybrid_switch_cfg(args.width, args.height, device, "NONE", batch_size=1)
    model = get_model(cfg, 19)
    fill_state_dict(model, 1, 3.0)
    model.eval()
    # the 8 frames as the evaluation loader delivers them with TEST batch size 8: ONE launch per layer over M = 8 x 8385 pixels
    # (eval-mode BatchNorm is per pixel: the result per frame is the one-at-a-time result; test_evaluate_path_matches_oracle).
    # `--eval-batch 1` walks them one at a time, as round 2 timed it.
    per = args.eval_batch
    frames = [{k: v.to(device) for k, v in synth_batch(per, args.height, args.width, seed=4000 + i).items()} for i in range(8 // per)]
    hist = torch.zeros(19, 19, dtype=torch.int64, device=device)

    def step(_i):
        with torch.no_grad():
            for f in frames:
                f = fresh(f)
                ops.upsample_argmax_hist(model(f["image"])[1]["out"], f["label"], hist, 19)
        return hist

    dt, _ = timed_loop(step, args.warmup, args.steps, device)
    roof = None if args.no_roofline else measure_roofline(step, 0, record=(rank == 0))
    gf = CONV_GFLOP.get((args.height, args.width))
    cfg_out = {"workload": f"forward-only evaluation of 8 frames {args.width}x{args.height} in batches of {per} (eval-mode forward + "
                           f"fused upsample/argmax/confusion matrix), DeepLabV2-ResNet50 ProDA head, random-init weights",
               "baseline_config": 1, "global_batch": 8 * world, "parallelism": f"dp{world} (replicas: no exchange)",
               "conv_tflop_per_step_per_gpu": 8 * gf[0] / 1e3 if gf else None}
    return {"metric": f"forward-only frames/sec {args.height}x{args.width}", "value": round(world * 8 * args.steps / dt, 4),
            "unit": "images/s", "dt": dt, "scaling": "weak", "config": cfg_out, "roofline": roof}


def other_configs(args, device):
    """Short driver-timed runs (1 warm-up + 3 timed passes each, no instrumentation) of the BASELINE configurations the
    default line does not time: config 1 (forward-only, 8 frames), config 2 (supervised step), config 5 (the adaptation step
    at 1024x2048) and config 3 on the STATIC side of the switch."""
    import copy
    import gc
    out = {}
    plan = [("config1_forward_only_8_frames", dict(config=1, height=512, width=1024), run_forward_only),
            ("config2_supervised_step", dict(config=2, height=512, width=1024), run_segmentation),
            ("config3_static_branch", dict(config=3, height=512, width=1024, branch="static"), run_adaptation),
            ("config5_adaptation_1024x2048", dict(config=5, height=1024, width=2048), run_adaptation),
            # the anchor of config 4's strong-scaling curve: the fixed global batch of 32 on ONE GPU (8 micro-batches per optimizer
            # step through step_sharded: per-micro-batch BatchNorm statistics, one gradient sum, one SGD step)
            ("config4_global_batch_32_on_one_gpu", dict(config=3, height=512, width=1024, global_batch=32, steps=2), run_adaptation)]
    for name, over, run in plan:
        sub = copy.copy(args)
        sub.warmup, sub.steps, sub.no_roofline, sub.no_exact_f32, sub.global_batch = 1, 3, True, True, 0
        sub.h2d, sub.no_h2d_leg = False, True
        for k, v in over.items():
            setattr(sub, k, v)
        try:
            res = run(sub, device, 0, 1)
            out[name] = {"value": res["value"], "unit": res["unit"], "ms_per_step": round(res["dt"] / sub.steps * 1e3, 3),
                         "steps": sub.steps, "warmup": sub.warmup, "metric": res["metric"]}
            if "branch" in res["config"]:
                out[name]["branch"] = res["config"]["branch"]
        except Exception as exc:  # a side measurement must never take the headline line down with it
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this parent never
    touches the GPU: no exec, no re-launch behind a HIP call), one per device, rendezvous on the loopback interface,
    rank 0's JSON line relayed on stdout.  Any failure -- fewer devices than ranks, a rank that dies -- is a non-zero exit:
    a line that says `n_gpus: 1` is never printed for `--gpus 8`."""
    n = args.gpus
    forced = "ONDA_FORCE_DEVICE" in os.environ  # test hook: several ranks on one device (over gloo)
    have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if not forced and have < n:
        raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s); refusing to print a line for fewer ranks than asked")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    sink = tempfile.TemporaryFile()  # rank 0's stdout: a file, so that nobody has to drain a pipe while the ranks are watched
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=sink if r == 0 else subprocess.DEVNULL))
    codes = watch_ranks(procs)
    sink.seek(0)
    text = sink.read().decode()
    sys.stdout.write(text)
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py --gpus {n}: rank exit codes {codes}")
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if not lines or json.loads(lines[-1]).get("n_gpus") != n:
        raise SystemExit(f"bench.py --gpus {n}: rank 0 did not report {n} ranks")


def watch_ranks(procs, poll_s=0.2, grace_s=5.0):
    """Wait for the self-started ranks.  A rank that ends with a non-zero code while others are still running would
    leave them inside a rendezvous or a collective until the distributed timeout (tens of minutes): the first such exit
    stops the others -- these exact children, by handle: terminate, then kill after `grace_s` -- and the caller exits
    non-zero within seconds.  Returns the exit codes."""
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            return codes
        if any(c not in (None, 0) for c in codes):
            live = [p for p in procs if p.poll() is None]
            for p in live:
                p.terminate()
            deadline = time.monotonic() + grace_s
            for p in live:
                try:
                    p.wait(timeout=max(0.0, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            return [p.returncode for p in procs]
        time.sleep(poll_s)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    if os.environ.get("ONDA_BENCH_FAIL_RANK") == os.environ.get("RANK", ""):
        raise SystemExit(7)  # test hook: this rank dies at start-up (tests/test_host_logic.py: the launcher must notice)
    point_miopen_at_tree()
    if args.eager_only:  # child of eager_rocm_boxed: this leg alone, its result as one JSON line
        torch.cuda.set_device(0)
        print(json.dumps(eager_rocm(args, "cuda:0")), flush=True)
        return
    from onda_amd import dist as odist
    rank, world, local = odist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    lib = library_identity()
    if not lib["fresh"]:
        raise SystemExit(f"libonda_hip.so was not compiled from the sources beside it: {lib}; run `python -m onda_amd.build`")
    run = {1: run_forward_only, 2: run_segmentation, 3: run_adaptation, 5: run_adaptation}[args.config]
    res = run(args, device, rank, world)
    dt = res.pop("dt")
    cpu = eager = others = eager_note = None
    headline = args.config in (3, 5) and not args.global_batch
    default_line = headline and args.config == 3 and (args.height, args.width) == (512, 1024) and args.batch == 4
    if rank == 0 and world == 1 and headline:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if default_line and not args.no_other_configs:
            others = other_configs(args, device)
        # the PyTorch-ROCm eager comparison runs LIVE whenever MIOpen's search results for this workload are in the tree
        # (seconds); without them only on request (--eager: the search takes ~10 minutes and fills tools/miopen_db)
        if default_line and not args.no_eager and args.eager:
            eager = eager_rocm(args, device)        # the tuning run: in this process, however long MIOpen searches
            gc.collect()
            torch.cuda.empty_cache()
        elif default_line and not args.no_eager and miopen_db_ready():
            eager, eager_note = eager_rocm_boxed(args)  # the driver's run: a child process with a time budget
        if default_line and not args.no_exchange_probe and not args.no_roofline:
            res["config"]["multi_gpu_exchange_on_one_gpu"] = exchange_probe(args, dt / args.steps * 1e3)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args)
    if rank == 0:
        dtype, note = arithmetic()
        res["config"].update({"conv_mode": note, "conv_accuracy": conv_accuracy_probe(device), "library": lib})
        vs = None
        if others:
            res["config"]["other_configs"] = others
        if eager:
            eager["speedup_of_this_repo"] = round(eager["ms_per_step"] / (dt / args.steps * 1e3), 3)
            res["config"]["eager_rocm"] = eager
            # BASELINE.md publishes no number for this metric; the reference point the north star names (">= 5x the
            # reference single-GPU PyTorch step") is measured in this same run, on this same GPU
            vs = round(res["value"] / eager["images_per_s"], 3)
            res["config"]["vs_baseline_is"] = ("value / config.eager_rocm.images_per_s: the same step on PyTorch-ROCm eager (MIOpen "
                                               "fp32, tuned), timed live in this run on this GPU (BASELINE.md holds no published number)")
        elif default_line:
            if eager_note:
                res["config"]["eager_rocm_skipped"] = eager_note
            # no tuned MIOpen find-db in the tree and no --eager: the committed measurement of the same step (same GPU
            # model, same workload, builder-run) is quoted with its source; vs_baseline stays null
            committed = {"ms_per_step": 302.9, "source": "profiles/r01_b_eager_pytorch_rocm.txt (tools/eager_baseline.py: MIOpen fp32, "
                         "cudnn.benchmark=True; with the reference's own flags -- benchmark off, deterministic on -- 11.3 s per step)"}
            committed["speedup_of_this_repo"] = round(committed["ms_per_step"] / (dt / args.steps * 1e3), 3)
            res["config"]["eager_rocm_committed"] = committed
        line = {"metric": res["metric"], "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": res["scaling"], "vs_baseline": vs, "dtype": dtype, "data": "synthetic", "config": res["config"],
                "roofline": res["roofline"], "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if odist.is_on():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
