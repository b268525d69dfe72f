"""Decode time of one VAE tile at the latent tile shapes the tiling policy produces (A/B of kernel-dispatch thresholds):
(5,64,96) = 768x512 clips, (6,52,84) = the 1280x768 spatial tiles, smaller ones for 512x512 / 256x256."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from vae_bench import synthetic_vae
vae = synthetic_vae("cuda:0")
for shape in ((5, 64, 96), (6, 52, 84), (5, 64, 64), (5, 32, 48), (5, 32, 32)):
    z = torch.randn(1, 16, *shape, device="cuda")
    vae._decode_tile(z); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        vae._decode_tile(z)
    torch.cuda.synchronize()
    print(shape, f"{(time.perf_counter() - t0) / 3 * 1e3:.1f} ms", flush=True)
