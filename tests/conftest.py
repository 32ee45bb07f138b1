import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CPU reference legs are small; on a many-core host (the GPU box has 256) the default
# one-thread-per-core OpenMP pool makes every tiny op pay a 256-way fork/join.
torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not config.pluginmanager.hasplugin("timeout"):  # (the marker stays legal where pytest-timeout is absent)
        config.addinivalue_line("markers", "timeout(seconds): per-test time limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    # a test that hangs (a rendezvous that never completes on some box, a kernel that never returns) must end as ONE failed
    # test after ten minutes, not stall the whole run: pytest-timeout is part of the image
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600))
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def digest(t, n=256):
    """Same (sum, abs-sum, strided sample) as tests/golden/make_golden.py."""
    f = t.detach().double().cpu().reshape(-1)
    idx = (torch.arange(n, dtype=torch.int64) * f.numel()) // n
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[idx].numpy()])


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
