"""`framework.dataset.segmentation_db` with the resize / normalise / label work moved to the GPU.

Mirror of the reference's `Segmentation_db` (framework/dataset/segmentation_db.py:16-99) for the
fields the adaptation step reads.  The constructor takes the same arguments; the difference is
WHERE the per-sample arithmetic runs:

  * `__getitem__` (a DataLoader worker process, as in the reference) only decodes the PNGs with PIL
    and returns the raw uint8 frames (`image_raw` u8[H0,W0,3] RGB, `label_ids` u8[H0,W0]);
  * `gpu_collate(samples)` (the training process) uploads them and runs
    `onda_amd.pipeline.GpuPreprocessor` -- Pillow's BICUBIC resize, BGR flip, ToTensor + Normalize,
    the two NEAREST label resizes and the id map, bit-identical to the reference's CPU path --
    returning the reference's batch dict: `image`, `label`, `label_res`, `image_path`, `label_path`.

Use: `DataLoader(ds, batch_size=4, num_workers=7, collate_fn=lambda s: s)` and call
`ds.gpu_collate(samples)` on what it yields (a GPU context must not be created inside workers).
Not mirrored (out of scope): RGB-coded label maps, `soft_predictions` caching,
`original_label`.
"""
from os import path

import numpy as np
import torch
from PIL import Image

from ...pipeline import GpuPreprocessor


class Segmentation_db(torch.utils.data.Dataset):
    def __init__(self, root_folder, pandas_metadata, class_map, image_size, labels_size=None, mean=(0.0, 0.0, 0.0),
                 std=(255.0, 255.0, 255.0), device="cuda:0"):
        """mean / std in [0, 255] (what the reference passes to `base_transform(mean, std)`,
        train_ouda.py:100-111); class_map: dict source id -> train id (`color_mapper`, rgb False)."""
        self.metadata = pandas_metadata
        self.root = root_folder
        self.image_size = image_size
        self.labels_size = labels_size if labels_size is not None else image_size
        if isinstance(class_map, dict):
            if isinstance(next(iter(class_map.keys())), (tuple, list)):
                raise NotImplementedError("onda_amd: RGB-coded label maps are not mirrored")
            lut = np.zeros(256, np.int64)
            for source, target in class_map.items():
                lut[source] = target
        else:
            lut = np.asarray(class_map)
        self._args = (tuple(self.image_size), tuple(self.labels_size), tuple(mean), tuple(std), lut, device)
        self._pre = None  # built lazily in the process that owns the GPU

    def __len__(self):
        return len(self.metadata)

    def __getitem__(self, index):
        row = self.metadata.iloc[index]
        image_path = path.join(self.root, row["image_path"])
        out = {"image_raw": torch.from_numpy(np.asarray(Image.open(image_path).convert("RGB"), np.uint8).copy()),
               "image_path": image_path}
        if "label_path" in row.keys() and row["label_path"] is not None:
            label_path = path.join(self.root, row["label_path"])
            out["label_ids"] = torch.from_numpy(np.asarray(Image.open(label_path), np.uint8).copy())
            out["label_path"] = label_path
        return out

    def gpu_collate(self, samples):
        if self._pre is None:
            size, lsize, mean, std, lut, device = self._args
            self._pre = GpuPreprocessor(size, lsize, mean, std, lut, device)
        labs = [s["label_ids"] for s in samples] if all("label_ids" in s for s in samples) else None
        batch = self._pre.batch([s["image_raw"] for s in samples], labs)
        batch["image_path"] = [s["image_path"] for s in samples]
        if labs is not None:
            batch["label_path"] = [s["label_path"] for s in samples]
        return batch
