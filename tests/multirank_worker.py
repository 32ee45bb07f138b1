"""Worker for tests/test_multirank_gpu.py: N ranks (one GPU shared over gloo, or one GPU each over
RCCL) run two adaptation steps on different micro-batches; replicas must stay identical."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from onda_amd import dist as odist  # noqa: E402


def main():
    rank, world, local = odist.init_from_env()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    from onda_amd.config import hybrid_switch_cfg
    from onda_amd.framework.domain_adaptation.methods.adaptation_model import switch_batch_statistics
    from onda_amd.framework.handlers import get_adapt_method, get_model
    from onda_amd.synthetic import fill_state_dict, synth_batch
    with tempfile.TemporaryDirectory() as tmp:
        cfg, spec = hybrid_switch_cfg(128, 64, dev, tmp, batch_size=2)
        model = get_model(cfg, 19)
        fill_state_dict(model, 1, 3.0)
        da = get_adapt_method(cfg)(model, cfg, spec)
        src = [synth_batch(2, 64, 128, seed=100 + 7 * rank + i) for i in range(2)]
        trg = [synth_batch(2, 64, 128, seed=200 + 7 * rank + i) for i in range(2)]
        torch.manual_seed(5)  # same dropout draws on every rank are not required, same prototypes are
        da.update_dynamic()
        switch_batch_statistics(da.model, False)
        # rank-specific batches (sharded loader): the class statistics are summed over ranks before the append
        da.calculate_prototypes([synth_batch(2, 64, 128, seed=50 + 3 * rank), synth_batch(2, 64, 128, seed=51 + 3 * rank)],
                                save=False)
        switch_batch_statistics(da.model, True)
        da.optimizer.zero_grad()
        for s in range(2):
            da.adjust_learning_rate(s, 4)
            log = da.step([src[s]], trg[s])
            da.update_ema()
        vals = []
        for mod in (da.model, da.ema_model):
            for p in mod.parameters():
                vals.append(p.detach().double().sum())
            for b in mod.buffers():
                vals.append(b.detach().double().sum())
        vals.append(da.prototypes.prototypes.double().sum())
        vals.append(da.prototypes.squared_mean.double().sum())
        vals.append(da.prototypes.counter.double().sum())
        vals.append(torch.tensor(float(da.prototypes.tau), device=dev, dtype=torch.float64))
        vals.append(torch.tensor(float(da.model_select.current), device=dev, dtype=torch.float64))
        mine = torch.stack(vals)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        loss = float(log["Total target loss"].detach())
    if rank == 0:
        print(f"MULTIRANK world={world} replicas_identical={same} loss={loss:.5f}", flush=True)
    dist.destroy_process_group()
    if not same:
        sys.exit(3)


if __name__ == "__main__":
    main()
