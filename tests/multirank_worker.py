"""Worker for tests/test_multirank_gpu.py.

* launched by torchrun with N ranks (one GPU shared over gloo, or one GPU each over RCCL): every rank runs adaptation
  steps on its own micro-batches; replicas must stay identical;
* launched as a plain process with ONDA_MR_SHARDS=N: ONE process runs the same micro-batches through
  ``step_sharded`` -- the sequential emulation of N ranks (rank-local batch statistics, averaged gradients, summed
  prototype statistics, averaged monitor scalars and running statistics).
Both write the state after every step to ONDA_MR_OUT (npz) so that the test can compare them.  Dropout2d is switched
off in both (the two layouts draw their masks from different streams; the exchange logic is what is under test)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from onda_amd import dist as odist  # noqa: E402

STEPS = 2


def digest_state(da):
    vals = {}
    for who, mod in (("student", da.model), ("teacher", da.ema_model)):
        flat = torch.cat([p.detach().reshape(-1).double() for p in mod.parameters()] +
                         [b.detach().reshape(-1).double() for b in mod.buffers()])
        idx = (torch.arange(20000, device=flat.device) * flat.numel()) // 20000
        vals[who] = flat[idx].cpu().numpy()
    vals["proto"] = da.prototypes.prototypes.double().cpu().numpy()
    vals["sqmean"] = da.prototypes.squared_mean.double().cpu().numpy()
    vals["counter"] = da.prototypes.counter.double().cpu().numpy()
    mon = da.intensity_ma.avg()
    vals["monitor"] = np.array([mon[k] for k in sorted(mon)])
    vals["switch"] = np.array([float(da.model_select.current), float(da.prototypes.tau)])
    return vals


def main():
    shards = int(os.environ.get("ONDA_MR_SHARDS", "0"))
    rank, world, local = odist.init_from_env()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.synthetic import fill_state_dict, synth_batch
    n = shards or world
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        cfg, spec = hybrid_switch_cfg(128, 64, dev, tmp, batch_size=2)
        model = get_model(cfg, 19)
        fill_state_dict(model, 1, 3.0)
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        da = get_adapt_method(cfg)(model, cfg, spec)
        mine = range(n) if shards else [rank]
        src = {r: [synth_batch(2, 64, 128, seed=100 + 7 * r + i) for i in range(STEPS)] for r in mine}
        trg = {r: [synth_batch(2, 64, 128, seed=200 + 7 * r + i) for i in range(STEPS)] for r in mine}
        torch.manual_seed(5)
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        # rank-specific batches (sharded loader): the class statistics are summed over ranks before the append
        proto_batches = [[synth_batch(2, 64, 128, seed=50 + 3 * r), synth_batch(2, 64, 128, seed=51 + 3 * r)] for r in mine]
        if shards:  # one process: batch j of every "rank" goes into append j
            for j in range(2):
                stats = None
                with torch.no_grad():
                    for r in range(n):
                        b = proto_batches[r][j]
                        pred = da.model(b["image"].to(dev))[1]
                        cls = torch.nn.functional.interpolate(b["label"].unsqueeze(1).float(), size=tuple(pred["out"].shape[2:])).view(-1)
                        flat, K, C = da.prototypes.class_statistics(pred["feat"], pred["out"].shape[1], classes=cls)
                        stats = flat if stats is None else stats + flat
                da.prototypes.append_from_statistics(stats, K, C)
        else:
            da.calculate_prototypes(proto_batches[0], save=False)
        switch_batch_statistics(da.model, True)
        da.optimizer.zero_grad()
        for k, v in digest_state(da).items():
            out[f"init_{k}"] = v
        for s in range(STEPS):
            da.adjust_learning_rate(s, 4)
            if shards:
                log = da.step_sharded([([src[r][s]], trg[r][s]) for r in range(n)])
            else:
                log = da.step([src[rank][s]], trg[rank][s])
            da.update_ema()
            for k, v in digest_state(da).items():
                out[f"s{s}_{k}"] = v
        loss = float(log["Total target loss"].detach())
        same = True
        if not shards and world > 1:
            mine_t = torch.cat([torch.from_numpy(out[f"s{STEPS - 1}_{k}"]).reshape(-1) for k in ("student", "teacher", "proto", "sqmean", "counter", "monitor", "switch")]).to(dev)
            gathered = [torch.empty_like(mine_t) for _ in range(world)]
            dist.all_gather(gathered, mine_t)
            same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        if os.environ.get("ONDA_MR_OUT"):
            np.savez(os.environ["ONDA_MR_OUT"], **out)
        print(f"MULTIRANK world={world} shards={shards} replicas_identical={same} loss={loss:.5f}", flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if not same:
        sys.exit(3)


if __name__ == "__main__":
    main()
