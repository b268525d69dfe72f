"""Full-size HunyuanVideo VAE decode on the engine: one (5,64,96) latent tile and the whole 5 s 768x512 clip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky.models.vae import AutoencoderKLHunyuanVideo

def synthetic_vae(device="cuda:0", seed=0):
    with torch.device("meta"):
        m = AutoencoderKLHunyuanVideo()
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, p in m.state_dict().items():
        if "norm" in k and k.endswith("weight"):
            sd[k] = torch.ones(p.shape, device=device)
        elif k.endswith("bias"):
            sd[k] = torch.zeros(p.shape, device=device)
        else:
            fan = p[0].numel()
            sd[k] = (torch.randn(p.shape, device=device, generator=g) / fan ** 0.5).half()
    m.load_state_dict(sd, assign=True)
    return m

if __name__ == "__main__":
    dev = "cuda:0"
    vae = synthetic_vae(dev)
    z = torch.randn(1, 16, 5, 64, 96, device=dev)
    vae._decode_tile(z); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = vae._decode_tile(z); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"tile (5,64,96) -> {tuple(out.shape)}: {(t1 - t0) * 1e3:.1f} ms  ({118.8 / (t1 - t0):.0f} TFLOP/s at 118.8 TFLOP/tile)", flush=True)
    assert torch.isfinite(out.float()).all()
    z = torch.randn(1, 16, 31, 64, 96, device=dev)
    t0 = time.perf_counter(); out = vae.decode(z).sample; torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"clip (31,64,96) -> {tuple(out.shape)}: {(t1 - t0):.2f} s (14 temporal tiles + blends)", flush=True)
    from kandinsky.generation_utils import frames_to_uint8
    u8 = frames_to_uint8(out); torch.cuda.synchronize()          # one pass of k5_frames_to_uint8 (the pipeline's own conversion)
    print("uint8 video", tuple(u8.shape), "mem GB", torch.cuda.max_memory_allocated() / 2**30)
