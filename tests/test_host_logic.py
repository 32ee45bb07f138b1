"""CPU: host-side logic of the product and the C-ABI surface (no kernel is launched)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "onda_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(onda_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from onda_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 38
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/onda_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.onda_version()
    # 19 ints, pad, run_if, stat_split, plain_schedule + pad, pix_table, pix_stride
    assert ctypes.sizeof(_lib.OndaConv) == 19 * 4 + 4 + 8 + 8 + 8 + 8 + 8 and ctypes.sizeof(_lib.OndaSgdEntry) == 48
    assert ctypes.sizeof(_lib.OndaSwitchCfg) == 16 + 6 * 8
    assert lib.onda_conv_tiles_m(33540) == 263
    assert lib.onda_conv_tiles_mc(33540, 256) == 263 and lib.onda_conv_tiles_mc(33540, 64) == 263  # (256 x 128 fp32 tiles: off)


def test_missing_library_fails_loudly(monkeypatch):
    from onda_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libonda_hip.so")
    with pytest.raises(_lib.OndaLibraryError, match="no fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected():
    from onda_amd.framework.model.deeplabv2 import get_deeplab_v2
    m = get_deeplab_v2(19, True, [3, 4, 6, 3], "ProDA").eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 128))
    with pytest.raises(NotImplementedError):
        get_deeplab_v2(19, True, [3, 4, 6, 3], "normal")


def test_model_tree_matches_reference_spec(golden):
    """376 state_dict keys / shapes / dtypes in the reference's order; parameter groups with the
    reference's duplicate pattern (fixture G6 holds what the reference's own walk yields)."""
    from onda_amd.framework.model.deeplabv2 import get_deeplab_v2
    from oracle import model as omodel
    m = get_deeplab_v2(19, True, [3, 4, 6, 3], "ProDA")
    sd = m.state_dict()
    spec = omodel.state_spec()
    assert [k for k, _, _ in spec] == list(sd.keys())
    for k, shape, dt in spec:
        assert tuple(sd[k].shape) == tuple(shape) and sd[k].dtype == dt, k
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 58882400
    m.multi_level = False
    g = golden("g6_optimizer")
    name_of = {id(p): n for n, p in m.named_parameters()}
    groups = m.optim_parameters(1e-5)
    c0 = {}
    for p in groups[0]["params"]:
        c0[name_of[id(p)]] = c0.get(name_of[id(p)], 0) + 1
    assert c0 == json.loads(str(g["group0_json"]))
    assert [name_of[id(p)] for p in groups[1]["params"]] == json.loads(str(g["group1_json"]))
    assert groups[1]["lr"] == pytest.approx(1e-4)


def test_feature_grid_formula():
    from onda_amd.synthetic import feature_hw
    assert feature_hw(512, 1024) == (65, 129) and feature_hw(1024, 2048) == (129, 257)
    assert feature_hw(64, 128) == (9, 17)


def test_monitor_and_switch_golden(golden):
    from onda_amd.framework.utils.monitoring import Monitor
    from onda_amd.framework.domain_adaptation.methods.prototypes_hybrid_switch import model_select
    g = golden("g5_switch")
    mon = Monitor(200, 0.003, "hamming")
    sel = model_select(model_select.static, [0.83, 0.9], 0.0002)
    for i, v in enumerate(g["seq"]):
        mon.add({"prior static": torch.tensor(float(v), dtype=torch.float64) if i % 2 else float(v)})
        a, d = mon.avg("prior static"), mon.dev_avg("prior static")
        sel.evaluate(a, d)
        assert a == g["avg"][i] and mon.exp("prior static") == g["exp"][i] and d == g["dev"][i]
        assert sel.current == g["current"][i]
    mon.eval(); sel.eval()
    mon.add({"prior static": 0.0}); sel.evaluate(0.0, -1.0)
    assert len(mon.current_dict["prior static"]) == 200 and sel.current == g["current"][-1]
    assert mon.avg("nope") == 1 and mon.dev_avg("nope") == 0 and mon.exp("nope") == 1


def test_cfg_semantics():
    from onda_amd.config import Cfg, hybrid_switch_cfg, unset
    cfg, spec = hybrid_switch_cfg()
    assert spec.SOFT_LABELS == {} and unset(spec.AUTO_DYNAMIC) and not unset(spec.BN_POLICY)
    assert cfg.SCHEME.RESOLUTION == [1024, 512] and spec.MA_LAMBDA == 0.9995 and spec.set_ == (25,)
    c = Cfg()
    c.A.B = 1
    assert c.A.B == 1 and c.X.Y == {}


def test_lr_schedule_and_hist():
    from onda_amd.framework.utils.func import fast_hist, lr_poly, per_class_iu
    assert lr_poly(1e-5, 10, 100, 0) == 1e-5
    assert lr_poly(1.0, 50, 100, 0.9) == pytest.approx(0.5 ** 0.9)
    a, b = np.array([0, 1, 1, 255, 2]), np.array([0, 1, 2, 1, 2])
    h = fast_hist(a, b, 3)
    assert h.sum() == 4 and h[1, 2] == 1
    np.testing.assert_allclose(per_class_iu(h), [1.0, 0.5, 0.5], rtol=1e-9)


def test_prototype_handler_host_side(tmp_path):
    from onda_amd.framework.domain_adaptation.methods.prototype_handler import prototype_handler
    with pytest.raises(ValueError):
        prototype_handler(distance_metric="cosine")
    h = prototype_handler(0.9995, 1, 0.3, "mahalanobis", confidence_regularization_threshold={})
    assert h.prototypes == 0 and h.confidence_regularization_threshold == 1
    assert h.load(str(tmp_path / "missing.pickle")) is False
    h.prototypes, h.squared_mean, h.counter = torch.ones(19, 256), torch.ones(19, 256) * 2, torch.ones(19)
    h.save(str(tmp_path / "p.pickle"))
    h2 = prototype_handler()
    assert h2.load(str(tmp_path / "p.pickle")) and torch.equal(h2.squared_mean, h.squared_mean)
    x = torch.arange(2 * 3 * 4 * 5.0).reshape(2, 3, 4, 5)
    assert torch.equal(h.transform(x), x.permute(0, 2, 3, 1).reshape(-1, 3))


def test_dropin_aliases_reference_names():
    import sys
    import onda_amd.dropin as dropin
    saved = {k: v for k, v in sys.modules.items() if k == "framework" or k.startswith("framework.")}
    try:
        dropin.install()
        from framework.domain_adaptation.methods.prototypes_hybrid_switch import hybrid_proDA, model_select  # noqa
        from framework.handlers import get_adapt_method, get_model  # noqa
        from framework.model.deeplabv2 import get_deeplab_v2
        import onda_amd.framework.model.deeplabv2 as mine
        assert get_deeplab_v2 is mine.get_deeplab_v2
    finally:
        for k in [k for k in sys.modules if k == "framework" or k.startswith("framework.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_amax_constants_match_header():
    """ops.AMAX_SLOTS is the buffer size the kernels assume (ONDA_AMAX_FLOATS in include/onda_hip.h)."""
    import os
    import re
    from onda_amd import ops
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "onda_hip.h")).read()
    floats = int(re.search(r"#define ONDA_AMAX_FLOATS (\d+)", hdr).group(1))
    slots = int(re.search(r"#define ONDA_AMAX_SLOTS (\d+)", hdr).group(1))
    assert ops.AMAX_SLOTS == floats and floats % slots == 0 and (floats // slots) * 4 == 128 and slots == 64


def test_library_carries_the_hash_of_its_sources():
    """onda_version() ends in the sha256 of the sources the library was compiled from: a stale or foreign .so is detected."""
    from onda_amd import _lib, build
    built, src = build.check_fresh()
    assert built == src and len(src) == 16
    assert _lib.load().onda_version().decode().endswith("src=" + src)
    assert ctypes_sizeof_pack_entry() == 48


def ctypes_sizeof_pack_entry():
    import ctypes
    from onda_amd import _lib
    return ctypes.sizeof(_lib.OndaPackEntry)


def test_monitor_ring_buffer_and_packed_adds():
    """Monitor on a ring: the window rolls over, packed device adds equal single adds, freeze ignores everything."""
    from onda_amd.framework.utils.monitoring import Monitor
    a, b = Monitor(5, 0.1, "hamming"), Monitor(5, 0.1, "hamming")
    vals = [0.1 * i * (-1) ** i for i in range(13)]
    for i, v in enumerate(vals):
        a.add({"x": v, "y": 2 * v})
        b.add_device(["x", "y"], torch.tensor([v, 2 * v], dtype=torch.float64))
        window = vals[max(0, i - 4): i + 1]
        assert a.avg("x") == pytest.approx(float(np.median(window)))
        assert a.avg("x") == b.avg("x") and a.exp("y") == pytest.approx(b.exp("y")) and a.dev_avg("x") == pytest.approx(b.dev_avg("x"))
        assert a.current_dict["x"] == pytest.approx(window)
    assert a.dev_avg("x") != 0 and a.dev_avg("missing") == 0 and a.avg("missing") == 1 and a.exp("missing") == 1
    a.eval()
    before = a.avg("x")
    a.add({"x": 100.0})
    a.add_device(["x"], torch.tensor([100.0]))
    assert a.avg("x") == before
    a.train()
    a.add({"x": 1.0}, reset=True)
    assert a.avg("x") == 1.0 and a.exp("x") == 1.0


def test_step_log_resolves_its_monitor_entries_on_first_read():
    """_StepLog: eager entries are there at once and can be extended without touching the lazy part; any read of the
    mapping (lookup of a missing key, iteration, len, `in`, items, copy) fills the monitor entries in exactly once."""
    from onda_amd.framework.domain_adaptation.methods.prototypes import _StepLog
    calls = []

    def lazy():
        calls.append(1)
        return {"x confidence ma": 0.25, "dev avg prior static": 0.0}

    def fresh():
        del calls[:]
        log = _StepLog({"Total target loss": 1.5}, lazy)
        log["encoder_lr"] = 1e-4
        log.update({"buff_loss": 2.0})
        assert log["Total target loss"] == 1.5 and log["buff_loss"] == 2.0 and not calls
        return log
    log = fresh()
    assert log["x confidence ma"] == 0.25 and len(calls) == 1
    assert log["dev avg prior static"] == 0.0 and len(calls) == 1
    for read in (lambda m: len(m), lambda m: list(m), lambda m: "x confidence ma" in m, lambda m: dict(m.items()),
                 lambda m: m.get("x confidence ma"), lambda m: m.copy(), lambda m: list(m.keys()), lambda m: list(m.values())):
        log = fresh()
        read(log)
        assert len(calls) == 1 and dict.__len__(log) == 5
        read(log)
        assert len(calls) == 1
    log = fresh()
    assert log.get("missing", 7) == 7 and len(calls) == 1


def test_monitor_pending_transfers_are_a_fifo_that_add_device_never_waits_for():
    """Device transfers queue up; a query for a series no pending transfer feeds leaves them alone; any query that involves
    a pending series, and any host-side add, drains them in order (sample order per series is kept); a non-blocking drain
    (what add_device does) stops at the first transfer still in flight; NaN on a gated key means "no sample"."""
    from onda_amd.framework.utils import monitoring
    m = monitoring.Monitor(5, 0.1, "hamming")
    m.add({"prior static": 0.5, "model": 0.1})

    class _Event:
        waited = 0

        def __init__(self, done=True):
            self.done = done

        def query(self):
            return self.done

        def synchronize(self):
            _Event.waited += 1
    m._pending = [(["model", "prior"], torch.tensor([0.2, 0.3]), _Event())]
    assert m.avg("prior static") == 0.5 and m.dev_avg("prior static") == 0 and _Event.waited == 0 and len(m._pending) == 1
    assert m.avg("model") == pytest.approx(0.15) and _Event.waited == 1 and not m._pending
    assert m.current_dict["model"] == pytest.approx([0.1, 0.2]) and m.avg("prior") == pytest.approx(0.3)
    m._pending = [(["model"], torch.tensor([0.4]), _Event())]
    m.add({"model": 0.5})  # the pending sample enters its ring BEFORE the new one
    assert m.current_dict["model"] == pytest.approx([0.1, 0.2, 0.4, 0.5]) and _Event.waited == 2
    # two transfers, the older one still in flight: a non-blocking drain takes nothing (order!), a read takes both
    m._pending = [(["prior"], torch.tensor([0.6]), _Event(done=False)), (["prior"], torch.tensor([0.7]), _Event())]
    m._flush(block=False)
    assert len(m._pending) == 2 and _Event.waited == 2
    assert m.current_dict["prior"] == pytest.approx([0.3, 0.6, 0.7]) and not m._pending
    # gated key: NaN = the quantity did not exist this step (the dynamic prior while the switch is static)
    m.gated.add("prior dynamic")
    m._pending = [(["prior dynamic", "model"], torch.tensor([float("nan"), 0.9]), _Event())]
    assert "prior dynamic" not in m.avg() and m.current_dict["model"][-1] == pytest.approx(0.9)
    m.add_device(["prior dynamic"], torch.tensor([0.25]))
    assert m.avg("prior dynamic") == 0.25


def test_replay_buffer_mirror():
    """Buffer_db: batches of consecutive samples, queue / random replacement, add_from_batch, and the nearest-neighbour
    label resize with OpenCV's index rule (floor(dst * src / dst_size))."""
    from onda_amd.framework.dataset.buffer_db import Buffer_db, label_to_outputs
    rng = np.random.default_rng(0)
    db = [{"image": torch.full((3, 8, 16), float(i)), "label": rng.integers(0, 19, (8, 16)).astype(np.uint8), "name": f"s{i}"}
          for i in range(5)]
    buf = Buffer_db(db, batch_size=2)
    assert len(buf) == 5 and buf.type_dict["image"] is torch.Tensor and buf.buffer[0]["domain"] == "source"
    batch = next(buf)
    assert batch["image"].shape == (2, 3, 8, 16) and batch["stored_predictions"].shape == (2, 8, 16)
    assert [float(batch["image"][i, 0, 0, 0]) for i in range(2)] == [0.0, 1.0]
    seen = [float(b["image"][0, 0, 0, 0]) for b in buf.sequential()]
    assert sorted(seen) == [0.0, 1.0, 2.0, 3.0, 4.0]
    target = {"image": torch.full((2, 3, 8, 16), 9.0), "label": torch.zeros(2, 8, 16, dtype=torch.uint8),
              "stored_predictions": torch.ones(2, 8, 16, dtype=torch.int64), "name": ["t0", "t1"]}
    buf.add_from_batch(target, 1)
    assert len(buf) == 5 and float(buf.buffer[-1]["image"][0, 0, 0]) == 9.0 and buf.buffer[-1]["domain"] == "target"
    assert float(buf.buffer[0]["image"][0, 0, 0]) == 1.0  # the oldest sample left the queue
    assert isinstance(buf.buffer[-1]["label"], np.ndarray) and buf.buffer[-1]["name"] == "t1"
    with pytest.raises(NotImplementedError):
        buf.add({}, policy="lifo")
    lab = rng.integers(0, 19, (64, 128)).astype(np.uint8)
    small = label_to_outputs(lab)
    assert small.shape == (9, 17)
    for r in (0, 3, 8):
        for c in (0, 5, 16):
            assert small[r, c] == lab[int(np.floor(r * 64 / 9)), int(np.floor(c * 128 / 17))]


def test_logging_sink_reads_device_scalars_once():
    from onda_amd import logging as olog
    got = []
    olog.set_sink(got.append)
    try:
        olog.log({"a": torch.tensor(1.5), "b": 2, "c": torch.tensor([3.0])})
    finally:
        olog.set_sink(None)
    assert got == [{"a": 1.5, "b": 2, "c": 3.0}]


def test_switch_plans_of_the_prototype_methods():
    """_prior_plan of every method (how teacher/static and dynamic priors are mixed) against the reference's rules
    (prototypes.py:232-255, prototypes_hybrid_switch.py:57-75, prototypes_hswitch.py:27-84, prototypes_vswitch.py:20-89),
    evaluated on stub objects: no model, no GPU."""
    from types import SimpleNamespace
    from onda_amd.config import Cfg
    from onda_amd.framework.domain_adaptation.methods import prototypes, prototypes_hswitch, prototypes_hybrid_switch, prototypes_vswitch
    from onda_amd.framework.utils.monitoring import Monitor

    def stub(cls, conf, **spec):
        mon = Monitor(200, 0.003, "hamming")
        mon.add({"prior static": conf})
        obj = SimpleNamespace(cfg_spec=Cfg.from_dict(spec), intensity_ma=mon)
        return obj, (lambda: cls._prior_plan(obj))

    _, plan = stub(prototypes.online_proDA, 0.7, SWITCH_PRIOR_THRESH=0.8, DYNAMIC_LAMBDA=1)
    assert plan() == (0.0, 1)  # low static confidence: the dynamic prior replaces
    _, plan = stub(prototypes.online_proDA, 0.9, SWITCH_PRIOR_THRESH=0.8, DYNAMIC_LAMBDA=1)
    assert plan() == (1.0, 0.0)
    _, plan = stub(prototypes.online_proDA, 0.9, SWITCH_PRIOR_THRESH=0, DYNAMIC_LAMBDA=0.5)
    assert plan() == (1.0, 0.5)  # no threshold: both priors add up
    _, plan = stub(prototypes.online_proDA, 0.9, SWITCH_PRIOR_THRESH=0, DYNAMIC_LAMBDA=0)
    assert plan() == (1.0, 0.0)
    sel = prototypes_hybrid_switch.model_select(0, [0.83, 0.9], 2e-4)
    obj, plan = stub(prototypes_hybrid_switch.hybrid_proDA, 0.5, DYNAMIC_LAMBDA=1)
    obj.model_select = sel
    assert plan() == (0.0, 1) and sel.current == sel.dynamic
    obj, plan = stub(prototypes_hybrid_switch.hybrid_proDA, 0.95, DYNAMIC_LAMBDA=1)
    obj.model_select = sel
    assert plan() == (1.0, 0.0) and sel.current == sel.static
    obj, plan = stub(prototypes_hswitch.hswitch_proDA, 0.88, SOFT_TRANS=True, DYNAMIC_LAMBDA=1)
    share = 0.88 * 25.0 / 3 - 41.0 / 6
    assert plan() == pytest.approx((share, 1 - share)) and obj.intensity_ma.avg("percentage_static") == pytest.approx(share)
    _, plan = stub(prototypes_hswitch.hswitch_proDA, 0.99, SOFT_TRANS=True, DYNAMIC_LAMBDA=1)
    assert plan() == (1, 0.0)
    _, plan = stub(prototypes_hswitch.hswitch_proDA, 0.5, SOFT_TRANS=False, SWITCH_PRIOR_THRESH=0.86, DYNAMIC_LAMBDA=2)
    assert plan() == (0, 2)
    vsel = prototypes_vswitch.model_select(0, 2e-4)
    obj, plan = stub(prototypes_vswitch.vswitch_proDA, 0.9, DYNAMIC_LAMBDA=1)
    obj.model_select = vsel
    assert plan() == (1.0, 0.0)
    vsel.evaluate(-1e-3)
    assert plan() == (0.0, 1)


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus 8` on a node that does not show 8 GPUs: non-zero exit, no JSON line (never `n_gpus: 1`)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ONDA_FORCE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and "{" not in out.stdout and "refusing" in out.stderr


def test_bench_launcher_stops_the_job_when_a_rank_dies():
    """A rank that dies while its siblings wait for it (rendezvous, a collective) must end the whole job within seconds,
    not after the distributed timeout: rank 1 exits at start-up (ONDA_BENCH_FAIL_RANK), rank 0 waits in the rendezvous for
    it; `bench.py --gpus 2` has to return non-zero, print no line, and do so in well under 30 s."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ONDA_FORCE_DEVICE="0", ONDA_DIST_BACKEND="gloo", ONDA_BENCH_FAIL_RANK="1")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=120)
    took = time.monotonic() - t0
    assert out.returncode != 0 and "{" not in out.stdout and "rank exit codes" in out.stderr, out.stderr[-800:]
    assert took < 30, took


def test_watch_ranks_kills_only_its_own_children():
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    sleeper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
    dier = subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(5)"])
    t0 = time.monotonic()
    codes = bench.watch_ranks([sleeper, dier], poll_s=0.05, grace_s=2.0)
    assert time.monotonic() - t0 < 10
    assert codes[1] == 5 and codes[0] not in (0, None)
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(2)]
    assert bench.watch_ranks(ok, poll_s=0.05) == [0, 0]


def test_live_k_step_fractions_of_the_dilated_convolutions():
    """bench.py's `pipe.executed_tflops` rests on onda_conv_l2_live_fraction / onda_conv_wgrad_l2_live_fraction (host
    arithmetic that repeats the kernels' dead-tap tests): checked here against a direct count -- a filter tap of a tile is dead
    when every output row the tile touches maps to an input row outside the image -- on the ASPP branches' shapes."""
    from ctypes import byref
    from onda_amd._lib import OndaConv, query

    def desc(B, H, W, cin, cout, k, dil):
        return OndaConv(B, H, W, cin, H, W, cout, k, k, 1, dil, dil * (k - 1) // 2, cin, cout, 0, 1, H, W, 0, None, 0, 0)

    def direct(B, H, W, cout, dil, bm=256, resident=256):
        M = B * H * W
        tiles_m, tiles_n = -(-M // bm), -(-cout // 128)
        tiles = tiles_m * tiles_n
        rem = tiles % resident
        dp = tiles - rem  # (the long K loops of these shapes always take the balanced schedule: the remainder runs every tap)
        live_steps = rem * 9
        for t in range(dp):
            tm = t // tiles_n
            r0, r1 = tm * bm // W, (min(M, tm * bm + bm) - 1) // W
            live = sum(any(0 <= (r % H) + (tp // 3 - 1) * dil < H for r in range(r0, r1 + 1)) for tp in range(9))
            live_steps += max(live, 1)
        return live_steps / (tiles * 9)

    assert query("onda_conv_l2_live_fraction", byref(desc(4, 65, 129, 512, 512, 1, 1)), 0) == 1.0
    for dil in (6, 12, 18, 24):
        got = query("onda_conv_l2_live_fraction", byref(desc(4, 65, 129, 2048, 256, 3, dil)), 0)
        assert abs(got - direct(4, 65, 129, 256, dil)) < 1e-12, (dil, got)
        assert 0.75 < got < 1.0
        w = query("onda_conv_wgrad_l2_live_fraction", byref(desc(4, 65, 129, 2048, 256, 3, dil)), 7)
        assert 0.7 < w < got  # (the weight gradient skips per 32-pixel step: finer than per 256-row tile)


def test_limb_row_format_round_trip_on_the_host():
    """The operand format of the pre-split kernels as include/onda_hip.h documents it -- element (row r, channel c): first limb
    at f16 index r * 2 * ld + (c / 32) * 64 + c % 32, second limb 32 further, both of x * 2^e with max|x| * 2^e in [2^14, 2^15),
    the second one times 2^11 -- written out with numpy and decoded by ops.materialize (the decoder the GPU tests read the
    kernels' output with): 22 significant bits back."""
    import numpy as np
    from onda_amd import ops
    rng = np.random.default_rng(5)
    B, H, W, C = 2, 3, 5, 96
    x = (rng.standard_normal((B, H, W, C)) * np.exp(rng.standard_normal((B, H, W, C)) * 3)).astype(np.float32)
    amax = np.abs(x).max()
    e = 15 - int(np.frexp(amax)[1])
    xs = x.astype(np.float64) * 2.0 ** e
    l1 = xs.astype(np.float16)
    l2 = ((xs - l1.astype(np.float64)) * 2048.0).astype(np.float16)
    rows, ld = B * H * W, C
    flat = np.zeros(rows * 2 * ld, dtype=np.float16)
    r, c = np.meshgrid(np.arange(rows), np.arange(C), indexing="ij")
    at = r * 2 * ld + (c // 32) * 64 + c % 32
    flat[at.ravel()] = l1.reshape(rows, C).ravel()
    flat[at.ravel() + 32] = l2.reshape(rows, C).ravel()
    slot = torch.zeros(ops.AMAX_SLOTS)
    slot[0] = float(amax)
    t = ops.limb_only((B, H, W, C), "cpu", ops.Limbs(torch.from_numpy(flat), slot, ld, rows * ld))
    back = ops.materialize(t).numpy()
    assert np.abs(back - x).max() <= 2.0 ** -21 * amax  # (two 11-bit limbs below the maximum's exponent)
    assert np.abs(back - x).max() > 0                    # ... and not a pass-through of fp32 values
