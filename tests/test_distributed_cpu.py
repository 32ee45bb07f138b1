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


def _flat_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from onda_amd import dist as od
    od.init_from_env("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    net[3].weight.requires_grad_(False)  # a parameter that stays out
    sync = od.GradSync(net, tail_floats=6, bucket_floats=1000, skip=lambda name: name == "2.bias")
    assert sync.active and len(sync.buckets) >= 2
    views = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert "2.bias" not in views and "3.weight" not in views
    for step in range(2):
        sync.zero()
        x = torch.randn(16, 64) * (rank + 1)
        net(x).sum().backward()                 # first backward of the step: no exchange yet
        launched_before = len(sync._pending)
        sync.arm()
        net(x * 0.5).pow(2).sum().backward()    # last backward: buckets go out from the hooks
        launched_in_backward = len(sync._pending)
        sync.tail.copy_(torch.arange(6.0) * (rank + 1))
        local = {n: p.grad.clone() for n, p in net.named_parameters() if n in views}
        n = sync.finish()
        for name, p in net.named_parameters():  # gradients are still views of the flat buffer
            if name in views:
                assert p.grad.data_ptr() == views[name].data_ptr()
    out[rank] = (n, local, {k: v.grad.clone() for k, v in net.named_parameters() if k in views}, sync.tail.clone(), launched_before,
                 launched_in_backward, net[2].bias.grad.clone())
    dist.destroy_process_group()


def test_flat_bucketed_gradient_exchange_with_tail():
    """GradSync: .grad are views of one flat buffer, buckets are all-reduced from the hooks of the LAST backward pass,
    the tail rides with the last bucket (rank-SUM), gradients come back as the rank-MEAN, skipped parameters stay local."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_flat_worker, args=(2, port, out), nprocs=2, join=True)
    (n0, local0, avg0, tail0, before0, during0, skipped0), (n1, local1, avg1, tail1, _, _, skipped1) = out[0], out[1]
    assert n0 == n1 == 64 * 32 + 32 + 32 * 8 + 4 + 6  # 2.bias (skipped) and 3.weight (frozen) are not in the buffer
    assert before0 == 0 and during0 >= 1              # nothing before arm(); at least one bucket out during the backward
    for k in local0:
        assert torch.allclose(avg0[k], (local0[k] + local1[k]) / 2, rtol=1e-6, atol=1e-7), k
        assert torch.equal(avg0[k], avg1[k])
    assert torch.equal(tail0, torch.arange(6.0) * 3) and torch.equal(tail0, tail1)
    assert not torch.equal(skipped0, skipped1)


def _lost_grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from onda_amd import dist as od
    od.init_from_env("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    sync = od.GradSync(net, bucket_floats=1000)
    assert len(sync.buckets) >= 2
    result = {}
    for mode in ("none_before_step", "none_between_passes"):
        sync.zero()
        x = torch.randn(16, 64) * (rank + 1)
        if mode == "none_before_step":
            net.zero_grad(set_to_none=True)      # every .grad leaves the flat buffer: autograd will allocate fresh tensors
        net(x).sum().backward()
        if mode == "none_between_passes":
            net[0].weight.grad = None            # ... or one gradient disappears mid-step (its first-pass part is lost everywhere)
            net[3].bias.grad = net[3].bias.grad.clone()  # ... or is replaced by a tensor outside the buffer
        sync.arm()
        net(x * 0.5).pow(2).sum().backward()
        local = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        sync.finish()
        for name, p in net.named_parameters():   # back inside the flat buffer, at its own slot
            off, n = sync._slot[id(p)]
            assert p.grad.data_ptr() == sync.flat.data_ptr() + 4 * off, (mode, name)
        result[mode] = (local, {n: p.grad.detach().clone() for n, p in net.named_parameters()})
    # a second exchange over the same module detaches the first (hooks gone, gradients own their storage again)
    sync.close()
    assert not sync.active and all(p.grad.data_ptr() != 0 and p._post_accumulate_grad_hooks in (None, {}) or
                                   len(p._post_accumulate_grad_hooks) == 0 for p in net.parameters())
    out[rank] = result
    dist.destroy_process_group()


def test_gradient_that_left_the_flat_buffer_is_exchanged_correctly():
    """The advisor's finding: a gradient autograd re-allocated (``.grad`` was None, or replaced) used to be copied into its
    slot only AFTER the slot's bucket had been all-reduced -- the exchanged values were stale and replicas diverged.  Now
    the value is moved into the slot when the parameter signals (before its bucket can leave): every rank ends with the
    rank-mean of the gradients the ranks actually computed."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lost_grad_worker, args=(2, port, out), nprocs=2, join=True)
    for mode in ("none_before_step", "none_between_passes"):
        (local0, avg0), (local1, avg1) = out[0][mode], out[1][mode]
        for k in local0:
            assert torch.allclose(avg0[k], (local0[k] + local1[k]) / 2, rtol=1e-6, atol=1e-7), (mode, k)
            assert torch.equal(avg0[k], avg1[k]), (mode, k)


def _double_signal_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from onda_amd import dist as od
    od.init_from_env("gloo")
    torch.manual_seed(0)

    class InPlace(torch.autograd.Function):
        """What ops.Conv2dFn does with a weight: accumulate into .grad in place, signal, return None for the parameter."""

        @staticmethod
        def forward(ctx, x, w, sync):
            ctx.save_for_backward(x, w)
            ctx.sync, ctx.w = sync, w
            ctx.set_materialize_grads(False)
            return x @ w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            ctx.w.grad.add_(x.t() @ g)
            ctx.sync.grad_ready(ctx.w)  # first signal; autograd's post-accumulate hook fires too (second signal)
            return g @ w.t(), None, None

    ws = [torch.nn.Parameter(torch.randn(40, 40) * 0.1) for _ in range(6)]
    holder = torch.nn.ParameterList(ws)
    sync = od.GradSync(holder, bucket_floats=3000)  # two parameters per bucket
    assert [len(b[2]) for b in sync.buckets] == [2, 2, 2, 0]  # (the last bucket carries the tail only)
    sync.zero()
    x = torch.randn(8, 40) * (rank + 1)
    y = x
    sync.arm()
    for w in ws:
        y = InPlace.apply(y, w, sync)
    y.sum().backward()
    launched_early = sum(sync._launched)
    sync.finish()
    # this rank's own gradients, from plain autograd on copies (the flat buffer may already hold sums by now)
    refs = [w.detach().clone().requires_grad_(True) for w in ws]
    y = x
    for w in refs:
        y = y @ w
    y.sum().backward()
    out[rank] = ([w.grad.clone() for w in refs], [w.grad.clone() for w in ws], launched_early)
    dist.destroy_process_group()


def test_parameters_that_signal_twice_count_once():
    """A conv weight signals the exchange twice per pass (explicitly after its in-place accumulation, and through autograd's
    post-accumulate hook, which fires even though backward returns None for it).  Counted twice, a bucket would leave
    when half of its members are complete; every rank must end with the mean of the COMPLETE gradients."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_double_signal_worker, args=(2, port, out), nprocs=2, join=True)
    (l0, a0, early0), (l1, a1, _) = out[0], out[1]
    assert early0 == 3  # the three full buckets went out during the backward pass, each only when BOTH members were complete
    for i in range(6):
        assert torch.allclose(a0[i], (l0[i] + l1[i]) / 2, rtol=1e-6, atol=1e-7), i
        assert torch.equal(a0[i], a1[i])


def _world8_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from onda_amd import dist as od
    od.init_from_env("gloo")
    torch.set_num_threads(1)
    torch.manual_seed(0)  # same weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(48, 40), torch.nn.ReLU(), torch.nn.Linear(40, 24), torch.nn.ReLU(),
                              torch.nn.Linear(24, 8), torch.nn.Linear(8, 4))
    idle = torch.nn.Parameter(torch.zeros(5))  # a parameter that gets NO gradient in any pass (the unused-but-registered case)
    holder = torch.nn.ModuleList([net])
    holder.register_parameter("idle", idle)
    import copy
    twin = copy.deepcopy(net)  # this rank's own gradients come from a copy: a bucket that left early is summed IN PLACE
    sync = od.GradSync(holder, tail_floats=7, bucket_floats=200)
    sizes = [e - s for s, e, _ in sync.buckets]
    assert len(sync.buckets) >= 4 and sizes[-1] >= 7, sizes  # >= 3 full buckets + the one that carries the tail
    # the tail travels with the LAST bucket, behind the last gradients
    assert sync.buckets[-1][1] == sync.flat.numel() and sync.tail.data_ptr() == sync.flat.data_ptr() + 4 * (sync.flat.numel() - 7)
    g = torch.Generator().manual_seed(100 + rank)  # every rank its own micro-batch
    steps = []
    for step in range(2):
        sync.zero()
        x = torch.randn(12, 48, generator=g)
        net(x).sum().backward()                      # first backward pass of the step (source replay): nothing leaves
        assert not sync._pending
        sync.arm()
        net(x * 0.5).pow(2).sum().backward()         # last backward pass: full buckets leave from the hooks
        early = sum(sync._launched)
        sync.tail.copy_(torch.arange(7.0) + rank)
        twin.zero_grad()
        twin(x).sum().backward()
        twin(x * 0.5).pow(2).sum().backward()
        local = [p.grad.detach().clone() for p in twin.parameters()]
        n = sync.finish(mean=False)                  # rank SUMS: the division by the world size happens inside the SGD kernel
        sums = [p.grad.detach().clone() for p in net.parameters()]
        # what ReplaySGD does with them on the way in (onda_sgd_multi's grad_scale = 1 / world): one multiply per element
        scaled = [s * (1.0 / world) for s in sums]
        steps.append((n, early, local, sums, scaled, sync.tail.clone(), idle.grad.clone()))
    out[rank] = steps
    dist.destroy_process_group()


def test_gradient_exchange_with_eight_ranks():
    """BASELINE config 4's world size (8 ranks) through GradSync: several buckets that leave during the last backward pass, the
    tail in the last bucket, a parameter that never receives a gradient, rank SUMS scaled by 1 / world on the way into SGD
    (`finish(mean=False)` + `ReplaySGD.grad_scale`): every rank ends with the same buffer and with the mean of the eight
    ranks' own gradients."""
    world = 8
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world8_worker, args=(world, port, out), nprocs=world, join=True)
    nparams = 48 * 40 + 40 + 40 * 24 + 24 + 24 * 8 + 8 + 8 * 4 + 4 + 5
    for step in range(2):
        n0, early0, _, sums0, scaled0, tail0, idle0 = out[0][step]
        assert n0 == nparams + 7
        assert early0 >= 3                                         # at least three buckets overlapped with the backward pass
        assert torch.equal(tail0, torch.arange(7.0) * world + sum(range(world)))  # the tail comes back as the rank SUM
        assert torch.equal(idle0, torch.zeros(5))                  # no gradient anywhere: zeros exchanged, zeros back
        for r in range(1, world):
            for a, b in zip(sums0, out[r][step][3]):
                assert torch.equal(a, b), (step, r)               # replicas hold bit-identical buffers
            assert torch.equal(tail0, out[r][step][5])
        for i in range(len(sums0)):
            ref = sum(out[r][step][2][i].double() for r in range(world)) / world
            assert (scaled0[i].double() - ref).abs().max() <= 2e-6 * ref.abs().max(), (step, i)  # fp32 sums of eight terms
