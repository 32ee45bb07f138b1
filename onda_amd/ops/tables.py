"""Multi-tensor launches over device tables (pinned staging buffers): duplicate-replaying SGD, teacher EMA."""
import ctypes
import os
from ctypes import byref

import torch

from .. import _lib
from .._lib import OndaConv, OndaLimbOut, call, query
from . import _state
from ._state import BN_EPS, GN_EPS, GN_GROUPS, HEAD_PAD, STEM_K
from .core import _p, _stream


_TABLE_STAGES = {}


def _table(entries, struct, device):
    """The entry table of a multi-tensor launch on `device`.  A pageable host->device copy would stop the host until
    the stream has drained (measured: 39 ms behind the backward pass for the optimizer's table, 13 ms per weight
    repack): the bytes go through a small ring of pinned staging buffers with an asynchronous copy instead; a buffer is
    reused once the event recorded behind its copy has completed."""
    arr = (struct * len(entries))(*entries)
    raw = bytearray(bytes(arr))
    n = len(raw)
    host = torch.frombuffer(raw, dtype=torch.uint8)
    if torch.device(device).type != "cuda":
        return host.to(device)
    ring = _TABLE_STAGES.setdefault(str(device), [])
    slot = None
    for cand in ring:
        if cand[0].numel() >= n and cand[1].query():
            slot = cand
            break
    if slot is None:
        if len(ring) >= 16:  # (cannot happen with a few tables per step; bound the ring anyway)
            slot = max(ring, key=lambda c: c[0].numel())
            slot[1].synchronize()
            if slot[0].numel() < n:
                slot[0] = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        else:
            slot = [torch.empty(max(n, 1 << 14), dtype=torch.uint8, pin_memory=True), torch.cuda.Event()]
            ring.append(slot)
    slot[0][:n].copy_(host)
    dev = torch.empty(n, dtype=torch.uint8, device=device)
    dev.copy_(slot[0][:n], non_blocking=True)
    slot[1].record()
    return dev


def sgd_multi(items, momentum, weight_decay, grad_scale=1.0):
    """items: list of (param, grad, buf, lr, times, fresh).  One launch for all tensors; every gradient element is
    multiplied by `grad_scale` on the way in."""
    blk, ents, first = query("onda_multi_tensor_block"), [], 0
    for p, g, b, lr, times, fresh in items:
        ents.append(_lib.OndaSgdEntry(p.data_ptr(), g.data_ptr(), b.data_ptr(), p.numel(), float(lr), int(times), int(fresh), first))
        first += -(-p.numel() // blk)
    dev = items[0][0].device
    table = _table(ents, _lib.OndaSgdEntry, dev)
    call("onda_sgd_multi", _p(table), len(ents), float(momentum), float(weight_decay), float(grad_scale), first, _stream())
    for p, *_ in items:
        torch.autograd.graph.increment_version(p)


def ema_multi(items, cache=None):
    """items: list of (k, q, keep, blend): k = k*keep + q*blend, one launch.  `cache` (a dict the caller keeps next to a
    FIXED list of tensors): the device table is rebuilt only when an address or a factor changed -- at the end of a step
    nothing is queued behind the optimizer launch, so the ~1.5 ms of host work that 320 table entries cost were 1.5 ms of
    idle device per step (tools/trace_gaps.py)."""
    key = [(k.data_ptr(), q.data_ptr(), keep, blend) for k, q, keep, blend in items]
    if cache is not None and cache.get("key") == key:
        table, n, first = cache["table"], cache["n"], cache["blocks"]
    else:
        blk, ents, first = query("onda_multi_tensor_block"), [], 0
        for k, q, keep, blend in items:
            ents.append(_lib.OndaEmaEntry(k.data_ptr(), q.data_ptr(), k.numel(), float(keep), float(blend), first))
            first += -(-k.numel() // blk)
        table, n = _table(ents, _lib.OndaEmaEntry, items[0][0].device), len(ents)
        if cache is not None:
            cache.update(key=key, table=table, n=n, blocks=first)
    call("onda_ema_multi", _p(table), n, first, _stream())
    torch.autograd.graph.increment_version([k for k, *_ in items])

