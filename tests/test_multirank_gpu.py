"""GPU: the batch-sharded step with 2 ranks.  On a single-GPU box both ranks share cuda:0 and
exchange over gloo (ONDA_DIST_BACKEND / ONDA_FORCE_DEVICE test hooks of onda_amd.dist); the
collectives, their order and the replicated state are the same as over RCCL with one GPU each."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_stay_identical():
    env = dict(os.environ)
    multi = torch.cuda.device_count() >= 2
    if not multi:
        env.update(ONDA_DIST_BACKEND="gloo", ONDA_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "multirank_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("MULTIRANK")][-1]
    assert "world=2" in line and "replicas_identical=True" in line, line
    assert "nan" not in line.lower()
