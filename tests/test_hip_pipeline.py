"""Input pipeline: host tables on CPU, HIP kernels on the GPU, both against the oracle
(Pillow's algorithm restated, pinned by G9) and against G9 itself.  Everything here is bit-exact."""
import numpy as np
import pytest
import torch

from oracle import pipeline as opipe

SIZES = [(64, 32), (50, 23), (37, 31), (48, 100), (2048, 1024), (1024, 512), (1024, 129), (512, 65), (1914, 1052), (7, 7),
         (100, 3)]


@pytest.mark.parametrize("n_in,n_out", SIZES)
def test_host_tables_match_oracle(n_in, n_out):
    from onda_amd.pipeline import bicubic_tables, nearest_table
    b, k, ks = bicubic_tables(n_in, n_out)
    ob, ok = opipe.resample_coeffs(n_in, n_out)
    assert ks == ok.shape[1] and np.array_equal(b, ob) and np.array_equal(k, ok)
    assert np.array_equal(nearest_table(n_in, n_out), opipe.nearest_table(n_in, n_out))


@pytest.mark.gpu
def test_g9_on_gpu(golden):
    from onda_amd.pipeline import GpuPreprocessor
    g = golden("g9_pipeline")
    for n in range(int(g["ncases"])):
        W, H = (int(v) for v in g[f"size{n}"])
        pre = GpuPreprocessor((W, H), mean=g["mean"], std=g["std"], id_map=g["lut"])
        t = pre.image(torch.from_numpy(g[f"img{n}"]).cuda())
        assert np.array_equal(t.cpu().numpy(), g[f"tensor{n}"]), n
        full, res = pre.labels(torch.from_numpy(g[f"lab{n}"]).cuda())
        assert np.array_equal(full.cpu().numpy(), g[f"label{n}"]) and np.array_equal(res.cpu().numpy(), g[f"label_res{n}"]), n


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [((1024, 2048), (1024, 512)), ((1024, 2048), (2048, 1024)), ((375, 1242), (1024, 512)),
                                     ((97, 131), (128, 64)), ((512, 1024), (1024, 512))])
def test_full_size_against_oracle(src, dst):
    """Cityscapes frames (2048x1024) to the BASELINE resolutions, an odd-sized source, an upscale and the
    identity size (Pillow then skips the pass; an identity tap reproduces it)."""
    from onda_amd.pipeline import GpuPreprocessor
    rng = np.random.default_rng(src[0] + dst[0])
    img = rng.integers(0, 256, (*src, 3), dtype=np.uint8)
    lab = rng.integers(0, 34, src, dtype=np.uint8)
    lut = np.concatenate([rng.integers(0, 19, 34), np.full(222, 255)]).astype(np.uint8)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    pre = GpuPreprocessor(dst, mean=mean, std=std, id_map=lut)
    got = pre.image(torch.from_numpy(img).cuda()).cpu().numpy()
    assert np.array_equal(got, opipe.preprocess_image(img, dst, mean, std))
    full, res = pre.labels(torch.from_numpy(lab).cuda())
    efull, eres = opipe.labels(lab, dst, lut)
    assert np.array_equal(full.cpu().numpy(), efull) and np.array_equal(res.cpu().numpy(), eres)
    assert res.shape == (dst[1] // 8 + 1, dst[0] // 8 + 1)


@pytest.mark.gpu
def test_batch_and_errors():
    from onda_amd.pipeline import GpuPreprocessor
    pre = GpuPreprocessor((128, 64))
    frames = [torch.randint(0, 256, (100, 200, 3), dtype=torch.uint8) for _ in range(2)]
    labs = [torch.randint(0, 19, (100, 200), dtype=torch.uint8) for _ in range(2)]
    b = pre.batch(frames, labs)
    assert b["image"].shape == (2, 3, 64, 128) and b["label"].shape == (2, 64, 128) and b["label_res"].shape == (2, 9, 17)
    with pytest.raises(RuntimeError):
        pre.image(frames[0])  # host tensor: no CPU fallback
    with pytest.raises(RuntimeError):
        pre.image(frames[0].float().cuda())


@pytest.mark.gpu
def test_segmentation_db_mirror(tmp_path):
    """PNG files -> DataLoader (decode only) -> gpu_collate == the oracle's per-sample transform."""
    import pandas as pd
    from PIL import Image
    from onda_amd.framework.dataset.segmentation_db import Segmentation_db
    rng = np.random.default_rng(3)
    rows, raw = [], []
    for i in range(3):
        img = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
        lab = rng.integers(0, 34, (90, 160), dtype=np.uint8)
        Image.fromarray(img).save(tmp_path / f"i{i}.png")
        Image.fromarray(lab).save(tmp_path / f"l{i}.png")
        rows.append({"image_path": f"i{i}.png", "label_path": f"l{i}.png"})
        raw.append((img, lab))
    ids = {i: (i % 19 if i % 3 else 255) for i in range(34)}
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    ds = Segmentation_db(str(tmp_path), pd.DataFrame(rows), ids, (64, 32), mean=mean, std=std)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, num_workers=0, collate_fn=lambda s: s)
    batch = ds.gpu_collate(next(iter(loader)))
    lut = np.zeros(256, np.int64)
    for k, v in ids.items():
        lut[k] = v
    for i, (img, lab) in enumerate(raw):
        assert np.array_equal(batch["image"][i].cpu().numpy(), opipe.preprocess_image(img, (64, 32), mean, std))
        full, res = opipe.labels(lab, (64, 32), lut)
        assert np.array_equal(batch["label"][i].cpu().numpy(), full) and np.array_equal(batch["label_res"][i].cpu().numpy(), res)
    assert batch["image"].shape == (3, 3, 32, 64) and batch["label_res"].shape == (3, 5, 9)
