"""Shared by the CPU (oracle) and GPU (HIP path) replays of fixture G14 -- the outer domain loop the driver calls
(train_ouda.py:227-261 -> hybrid_proDA.train, prototypes.py:466-520): the run's constants (the generator's,
tests/golden/make_golden.py::G14), its synthetic loaders, and the comparison of a captured log stream with the
reference's."""
import json

import numpy as np

G14 = dict(head_scale=7.0, monitor=4, perc_fill=0.0009, np_seed=3, steps=3, val_frames=10, lr=2e-7)
# 2 prototype batches + 3 masks per step x 6 steps, drawn from torch's default generator after manual_seed(123)
N_MASKS = 2 + 3 * 2 * G14["steps"]


def loaders():
    from onda_amd.synthetic import ListLoader, synth_batch
    n = G14["steps"]
    src = ListLoader([synth_batch(2, 64, 128, seed=1400 + i) for i in range(2)])
    domains = [ListLoader([synth_batch(2, 64, 128, seed=1500 + 10 * d + i) for i in range(n)]) for d in range(2)]
    val = {"val": ListLoader([synth_batch(1, 64, 128, seed=1600 + i) for i in range(G14["val_frames"])])}
    return src, domains, val


def scalar(v):
    import torch
    if isinstance(v, torch.Tensor):
        return float(v.detach().double().cpu()) if v.numel() == 1 else None
    if isinstance(v, np.ndarray):
        return float(v) if v.ndim == 0 else None
    return float(v) if np.isscalar(v) else None


# what a deviation means depends on the entry: losses and confidences are smooth in the weights; the trend is a difference
# of near-equal levels; counts and mIoU move in steps of one pixel
def _close(key, mine, ref, step):
    if np.isnan(ref):
        return np.isnan(mine)
    grow = 1.0 + 0.5 * step  # (the fixture's learning rate keeps the six steps comparable: make_golden.py::g14_config)
    if key == "pseudolabel_pixel_num":
        return abs(mine - ref) <= 2
    if key == "output & prototype agreement":
        return abs(mine - ref) <= 3.0 / 306
    if key == "dev avg prior static":
        return abs(mine - ref) <= 2e-4 * grow
    if key.startswith("Val mIoU") or key.startswith("Val std IoU"):
        return abs(mine - ref) <= 2e-4 * grow
    if key == "Total buffer updates":
        return mine == ref
    return abs(mine - ref) <= 5e-3 * grow * max(abs(ref), 1e-3)


def compare_logs(g, logs, image_key=lambda entry: entry, worst=None):
    """`logs`: the captured dictionaries in emission order.  Scalars against the reference's with the tolerances above;
    sample class maps pixel for pixel up to a few near-tie pixels."""
    ref = json.loads(str(g["logs_json"]))
    assert len(logs) == len(ref) == 7
    for i, (mine, want) in enumerate(zip(logs, ref)):
        got = {k: scalar(v) for k, v in mine.items()}
        missing = [k for k in want if got.get(k) is None]
        assert not missing, (i, missing)
        extra = [k for k, v in got.items() if v is not None and k not in want]
        assert not extra, (i, extra)
        step = max(i - 1, 0)
        for k, v in want.items():
            ok = _close(k, got[k], v, step)
            if worst is not None and v == v and abs(v) > 0:
                worst[k] = max(worst.get(k, 0.0), abs(got[k] - v) / abs(v))
            assert ok, (i, k, got[k], v)
        maps = [k for k in g if k.startswith(f"log{i}_img_")]
        assert (len(maps) == G14["val_frames"]) == (i in (3, 6))
        for k in maps:
            entry = image_key(mine[k[len(f"log{i}_img_"):]])
            differ = (np.asarray(entry) != g[k]).mean()
            assert differ <= 2e-3 * (1 + step), (i, k, differ)
