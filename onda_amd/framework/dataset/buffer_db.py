"""Drop-in for ``framework/dataset/buffer_db.py`` (:32-125): the in-memory replay buffer the source loader of the
online methods is (``BUFFER_DYNAMIC``), and the target of ``online_proDA.buffer_update``.

Same surface: ``Buffer_db(initial_db, batch_size, domain, channels)``, iteration yields collated batches of
``batch_size`` consecutive samples (a fresh permutation every wrap), ``len``, ``[index]`` (one permuted sample as a
batch of one), ``sequential()``, ``add(item, policy)`` ("queue": drop the oldest; "random": overwrite one),
``add_from_batch(batch, index, domain)``, ``label_to_outputs(label)`` (the label at feature resolution, nearest
neighbour with OpenCV's index rule ``floor(dst * src / dst_size)`` -- numpy here, cv2 is not a dependency).
"""
from collections import deque

import numpy as np
import torch
from torch.utils.data import IterableDataset
from torch.utils.data.dataloader import default_collate


def label_to_outputs(label, channels=19):
    label = np.asarray(label)
    height, width = label.shape
    out_h, out_w = height // 8 + 1, width // 8 + 1
    rows = np.minimum((np.arange(out_h) * (height / out_h)).astype(np.int64), height - 1)
    cols = np.minimum((np.arange(out_w) * (width / out_w)).astype(np.int64), width - 1)
    return label[rows[:, None], cols[None, :]]


class Buffer_db(IterableDataset):
    def __init__(self, initial_db, batch_size, domain="source", channels=19):
        self.channels = channels
        self.batch_size = batch_size
        self.buffer = deque()
        for i in range(len(initial_db)):
            sample = dict(initial_db[i])
            sample["domain"] = domain
            sample["stored_predictions"] = sample["label"]
            self.buffer.append(sample)
        if not self.buffer:
            raise ValueError("Buffer_db needs a non-empty initial dataset")
        self.type_dict = {key: type(value) for key, value in self.buffer[0].items()}
        self.pos = 0
        self.permutation = np.random.permutation(len(self.buffer))

    def __len__(self):
        return len(self.buffer)

    def __iter__(self):
        return self

    def __next__(self):
        items = []
        for _ in range(self.batch_size):
            items.append(self.buffer[self.pos])
            self.pos = (self.pos + 1) % len(self)
            if self.pos == 0:
                self.permutation = np.random.permutation(len(self.buffer))
        return default_collate(items)

    def __getitem__(self, index):
        return default_collate([self.buffer[self.permutation[index]]])

    def sequential(self):
        for i in range(len(self)):
            yield self[i]

    def add(self, item, policy="queue"):
        if policy == "queue":
            self.buffer.popleft()
            self.buffer.append(item)
        elif policy == "random":
            self.buffer[np.random.randint(len(self.buffer))] = item
        else:
            raise NotImplementedError(f"the policy {policy}, has not been implemented")

    def add_from_batch(self, batch, index, domain="target"):
        batch["domain"] = domain
        item = {}
        for key, kind in self.type_dict.items():
            value = batch[key] if key == "domain" and isinstance(batch[key], str) else batch[key][index]
            if type(value) != kind and isinstance(value, torch.Tensor):
                value = value.cpu().numpy()
            item[key] = value
        self.add(item)
