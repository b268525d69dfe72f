"""VERDICT r3 next #7 premise check: how much of the timed VAE decode is HOST GLUE (tile slicing, blends, the final concatenation — what a
single `k5_vae_decode` C entry would remove) and how much is the decoder network itself?  The clip decode of bench.py (31 latent frames,
64 x 96: 14 temporal tiles) with a HIP-event pair around every `_decode_tile` engine call: glue = decode wall time - sum of the tile calls.
The 930 `__amd_rocclr_copyBuffer` calls of profiles/r03_vae_kernel_stats.md are counted here by WHERE they happen: torch's allocator / copy
statistics before and after the (already warm) decode.

    python tools/vae_glue_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from vae_bench import synthetic_vae

vae = synthetic_vae("cuda:0")
z = torch.randn(1, 16, 31, 64, 96, device="cuda:0")
vae.decode(z)            # warm: weights staged into the engine's packed layout (the copies), workspaces allocated
torch.cuda.synchronize()
events = []
orig = vae._decode_tile


def timed_tile(t):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = orig(t); b.record()
    events.append((a, b))
    return out


vae._decode_tile = timed_tile
for rep in range(3):
    events.clear()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s0.record(); out = vae.decode(z).sample; s1.record()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    tiles = sum(a.elapsed_time(b) for a, b in events)
    gpu = s0.elapsed_time(s1)
    print(f"decode of the 5 s clip: wall {wall:8.1f} ms, GPU span {gpu:8.1f} ms; {len(events)} tile calls {tiles:8.1f} ms; "
          f"everything else (slices, blends, concatenation, launch gaps between tiles) {gpu - tiles:6.1f} ms = {100 * (gpu - tiles) / gpu:.2f} %", flush=True)
