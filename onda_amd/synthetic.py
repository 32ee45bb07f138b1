"""Deterministic synthetic weights and batches for the adaptation hot path.

There is no dataset or checkpoint on the GPU box, so parity fixtures, the smoke
run and ``bench.py`` all draw their state from here.  Every tensor is produced
from a per-key ``torch.Generator`` on the CPU, so the result does not depend on
iteration order, on the module classes that own the keys, or on the device the
model is later moved to.  The same function fills the reference's
``ResNetMulti`` (``framework/model/deeplabv2.py:260-331``, when the golden
vectors are generated) and this repo's drop-in model.

Batch schema follows the reference loader (``framework/dataset/segmentation_db.py:56-80``):
``image f32[B,3,H,W]`` (already normalised), ``label u8[B,H,W]`` (255 = ignore),
``label_res u8[B,H/8+1,W/8+1]``.
"""
import math
import zlib

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(name: str, ref: torch.Tensor, seed: int, head_scale: float = 1.0) -> torch.Tensor:
    """Value for state_dict entry `name` (shape/dtype taken from `ref`)."""
    g = _gen(name, seed)
    shape = tuple(ref.shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=ref.dtype)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if len(shape) == 4:  # conv weight: He-style so that eval-mode activations stay O(1)
        fan_in = shape[1] * shape[2] * shape[3]
        w = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        if name.endswith("head.1.weight"):
            w = w * head_scale
        return w
    if len(shape) == 2:  # SE linears
        return torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[1])
    if len(shape) == 1 and leaf == "weight":  # BN / GN gamma
        gamma = 0.75 + 0.5 * torch.rand(shape, generator=g)
        if ".bn3." in name:  # keep the residual trunk from growing over 16 blocks
            gamma = gamma * 0.4
        return gamma
    if len(shape) == 1 and leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    raise ValueError(f"no synthetic rule for {name} {shape}")


@torch.no_grad()
def fill_state_dict(module: torch.nn.Module, seed: int = 1, head_scale: float = 1.0) -> None:
    """Overwrite every parameter and buffer of `module` in place."""
    for name, t in module.state_dict().items():
        v = synth_tensor(name, t, seed, head_scale).to(dtype=t.dtype)
        t.copy_(v.to(t.device))


def synth_batch(batch: int, height: int, width: int, seed: int = 7, num_classes: int = 19,
                ignore_frac: float = 0.1) -> dict:
    """One synthetic batch in the reference loader's dict schema (CPU tensors)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    image = torch.randn(batch, 3, height, width, generator=g)
    label = torch.randint(0, num_classes, (batch, height, width), generator=g, dtype=torch.int64)
    drop = torch.rand(batch, height, width, generator=g) < ignore_frac
    label[drop] = 255
    h, w = feature_hw(height, width)
    label_res = torch.randint(0, num_classes, (batch, h, w), generator=g, dtype=torch.int64)
    drop = torch.rand(batch, h, w, generator=g) < ignore_frac
    label_res[drop] = 255
    return {"image": image, "label": label.to(torch.uint8), "label_res": label_res.to(torch.uint8)}


def feature_hw(height: int, width: int):
    """Feature-grid size of the network for an H x W input (stem s2,
    ceil-mode pool s2, layer2 s2) = the loader's int(H/8+1) (segmentation_db.py:91)."""
    def one(n):
        n1 = (n - 1) // 2 + 1
        n2 = -(-(n1 - 1) // 2) + 1
        return (n2 - 1) // 2 + 1
    return one(height), one(width)


def synth_prototypes(feat_dim: int = 256, num_classes: int = 19, seed: int = 11):
    """A valid (prototypes, squared_mean, counter) state with the value range the
    reference's shipped pickle shows (values in about [-11, 11])."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    proto = 2.0 * torch.randn(num_classes, feat_dim, generator=g)
    spread = 0.5 + torch.rand(num_classes, feat_dim, generator=g)
    squared_mean = proto ** 2 + spread ** 2
    counter = torch.randint(200, 5000, (num_classes,), generator=g).float()
    return proto, squared_mean, counter


class ListLoader:
    """A list of batches with the loader surface the adaptation loop touches (``prototypes.py:466-520`` of the reference:
    ``len``, iteration restarted when it runs out, ``add_from_batch`` of the replay buffer, ``buffer_db.py``): the
    synthetic stand-in for a ``DataLoader`` / ``Buffer_db`` in the outer-loop fixture G14 and in the tests that replay
    it.  ``add_from_batch`` only RECORDS (index, the stored label map of that sample) -- the batches stay what they are,
    so a run is a function of the seeds alone."""

    def __init__(self, batches):
        self.batches = list(batches)
        self.added = []

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)

    def sequential(self):
        return iter(self.batches)

    def add_from_batch(self, batch, index):
        stored = batch["stored_predictions"]
        self.added.append((int(index), stored[index].detach().to("cpu", torch.uint8).clone()))
