"""Throughput of the GPU input pipeline on Cityscapes-sized frames (2048x1024 -> 1024x512)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from onda_amd.pipeline import GpuPreprocessor
pre = GpuPreprocessor((1024, 512), mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])
img = torch.randint(0, 256, (1024, 2048, 3), dtype=torch.uint8, device="cuda")
lab = torch.randint(0, 34, (1024, 2048), dtype=torch.uint8, device="cuda")
for _ in range(5):
    pre.image(img); pre.labels(lab)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    pre.image(img); pre.labels(lab)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
byts = img.numel() + 1024 * 1024 * 3 * 2 + 3 * 512 * 1024 * 4 + lab.numel() // 4 + 512 * 1024  # frame in, tmp out+in, tensor out, labels
print(f"{dt * 1e6:.1f} us per sample (image + labels), {1 / dt:.0f} samples/s, {byts / dt / 1e9:.0f} GB/s algorithmic")
