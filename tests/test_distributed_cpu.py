"""CPU, world_size 2, gloo: the three exchanges of the batch-sharded step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from onda_amd import dist as od
    rk, ws, _ = od.init_from_env("gloo")
    assert (rk, ws) == (rank, world) and od.is_on() and od.world_size() == 2
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.Linear(4, 2))
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    n = od.GradSync(net).all_reduce()
    grads = [p.grad.clone() for p in net.parameters()]
    # prototype statistics: [sum feat | sum feat^2 | count] summed before the blend
    stats = torch.arange(10.0) * (rank + 1)
    od.all_reduce_sum(stats)
    conf = od.all_reduce_mean(torch.tensor([0.8 + 0.1 * rank, 0.5]))
    mx = od.all_reduce_max(torch.tensor([float(rank)]))
    od.barrier()
    out[rank] = (n, grads, stats, conf, mx)
    dist.destroy_process_group()


def test_gradient_and_statistic_exchange():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][0] == 8 * 4 + 4 + 4 * 2 + 2
    for i in range(4):
        assert torch.equal(out[0][1][i], out[1][1][i])
        assert torch.allclose(out[0][1][i], torch.full_like(out[0][1][i], 1.5 * (i + 1)))
    assert torch.equal(out[0][2], torch.arange(10.0) * 3) and torch.equal(out[1][2], out[0][2])
    assert torch.allclose(out[0][3], torch.tensor([0.85, 0.5])) and torch.equal(out[0][3], out[1][3])
    assert out[0][4].item() == 1.0


def test_single_process_is_a_noop():
    from onda_amd import dist as od
    assert not od.is_on() and od.world_size() == 1 and od.rank() == 0
    t = torch.ones(3)
    assert od.all_reduce_sum(t) is t and od.all_reduce_mean(t) is t
    assert od.GradSync(torch.nn.Linear(2, 2)).all_reduce() == 0
