"""Full-size HunyuanVideo VAE encode on the engine: one tile and a 33-frame 768x512 clip (I2V conditioning path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from vae_bench import synthetic_vae

if __name__ == "__main__":
    dev = "cuda:0"
    vae = synthetic_vae(dev)
    x = torch.randn(1, 3, 17, 512, 768, device=dev)
    vae._encode_tile(x); torch.cuda.synchronize()
    t0 = time.perf_counter(); m = vae._encode_tile(x); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"tile (17,512,768) -> {tuple(m.shape)}: {(t1 - t0) * 1e3:.1f} ms", flush=True)
    assert torch.isfinite(m.float()).all()
    x = torch.randn(1, 3, 33, 512, 768, device=dev)
    t0 = time.perf_counter(); out = vae.encode(x).latent_dist.mode(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"clip (33,512,768) -> {tuple(out.shape)}: {(t1 - t0):.2f} s", flush=True)
