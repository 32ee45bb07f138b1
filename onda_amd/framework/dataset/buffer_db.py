"""Drop-in for ``framework/dataset/buffer_db.py`` (:32-125): the in-memory replay buffer that stands in for the source
loader of the online methods (``BUFFER_DYNAMIC``) and receives ``online_proDA.buffer_update``'s target samples.

Same surface and observable order as the reference -- ``Buffer_db(initial_db, batch_size, domain, channels)``, ``len``,
``next()`` = ``batch_size`` consecutive samples in age order starting where the last batch ended, ``[i]`` = the sample a
per-wrap random permutation maps ``i`` to (as a batch of one), ``sequential()``, ``add(item, policy)`` ("queue": the
oldest sample leaves; "random": one sample is overwritten), ``add_from_batch(batch, index, domain)``,
``label_to_outputs(label)`` -- on a different store: a fixed array of slots used as a ring.  Replacing the oldest
sample overwrites one slot and moves the ring's origin (the reference pops and appends a deque, shifting every index);
``buffer`` exposes the samples in age order for code that peeks at it.

Out of scope for the BASELINE configs (``BUFFER_DYNAMIC: False`` in all of them); kept because the step's
``buffer_update`` writes into it.
"""
import numpy as np
import torch
from torch.utils.data import IterableDataset
from torch.utils.data.dataloader import default_collate


def label_to_outputs(label, channels=19):
    """The label map at feature resolution (H // 8 + 1, W // 8 + 1): nearest neighbour with OpenCV's index rule
    ``src = floor(dst * src_size / dst_size)`` (the reference calls cv2.resize(..., INTER_NEAREST); numpy here)."""
    label = np.asarray(label)
    src = np.array(label.shape[:2])
    dst = src // 8 + 1
    pick = [np.minimum(np.floor(np.arange(d) * (s / d)).astype(np.int64), s - 1) for s, d in zip(src, dst)]
    return label[np.ix_(*pick)]


class _AgeOrder:
    """Read / write view of the ring in age order (index 0 = oldest, -1 = newest)."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return len(self._o._slots)

    def __getitem__(self, i):
        return self._o._slots[self._o._slot_of(i)]

    def __setitem__(self, i, item):
        self._o._slots[self._o._slot_of(i)] = item

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Buffer_db(IterableDataset):
    def __init__(self, initial_db, batch_size, domain="source", channels=19):
        self.channels, self.batch_size = channels, batch_size
        self._slots = [self._adopt(initial_db[i], domain) for i in range(len(initial_db))]
        if not self._slots:
            raise ValueError("Buffer_db needs a non-empty initial dataset")
        self._origin = 0  # slot of the oldest sample
        self.type_dict = {key: type(value) for key, value in self._slots[0].items()}
        self.pos = 0      # age index the next batch starts at
        self.permutation = np.random.permutation(len(self._slots))
        self.buffer = _AgeOrder(self)

    @staticmethod
    def _adopt(sample, domain):
        sample = dict(sample)
        sample.update(domain=domain, stored_predictions=sample["label"])
        return sample

    def _slot_of(self, age_index):
        n = len(self._slots)
        if not -n <= age_index < n:
            raise IndexError(age_index)
        return (self._origin + age_index) % n

    # ---- reading -----------------------------------------------------------------------------------------------------
    def __len__(self):
        return len(self._slots)

    def __iter__(self):
        return self

    def __next__(self):
        n = len(self._slots)
        ages = [(self.pos + k) % n for k in range(self.batch_size)]
        wrapped = self.pos + self.batch_size >= n
        self.pos = (self.pos + self.batch_size) % n
        batch = default_collate([self._slots[self._slot_of(a)] for a in ages])
        if wrapped:  # every pass over the buffer gets a new permutation for the indexed reads
            self.permutation = np.random.permutation(n)
        return batch

    def __getitem__(self, index):
        return default_collate([self._slots[self._slot_of(int(self.permutation[index]))]])

    def sequential(self):
        return (self[i] for i in range(len(self)))

    # ---- writing -----------------------------------------------------------------------------------------------------
    def add(self, item, policy="queue"):
        if policy == "queue":    # the oldest slot is overwritten and becomes the newest
            self._slots[self._origin] = item
            self._origin = (self._origin + 1) % len(self._slots)
        elif policy == "random":
            self._slots[np.random.randint(len(self._slots))] = item
        else:
            raise NotImplementedError(f"the policy {policy}, has not been implemented")

    def add_from_batch(self, batch, index, domain="target"):
        """Sample `index` of a collated batch, every field brought back to the type the buffer was built with."""
        item = {}
        for key, kind in self.type_dict.items():
            if key == "domain":
                item[key] = domain
                continue
            value = batch[key][index]
            if isinstance(value, torch.Tensor) and kind is not torch.Tensor:
                value = value.cpu().numpy()
            item[key] = value
        batch["domain"] = domain  # (the reference tags the batch as well)
        self.add(item)
