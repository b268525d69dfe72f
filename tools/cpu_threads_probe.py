import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import k5_oracle as O
cfg = O.DitConfig(**dict(O.LITE_2B, num_visual_blocks=1, num_text_blocks=0))
man = O.state_dict_manifest(cfg)
g = torch.Generator().manual_seed(0)
sd = {k: torch.randn(v, generator=g) * 0.02 for k, v in man.items() if k.startswith("visual_transformer_blocks.0.")}
n = 3328
x, text, temb = torch.randn(n, 1792, generator=g), torch.randn(256, 1792, generator=g), torch.randn(1, 512, generator=g)
cs = torch.ones(n, 32), torch.zeros(n, 32)
print("cores", os.cpu_count())
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    O.decoder_block(sd, "visual_transformer_blocks.0", x[:256], text, temb, cs[0][:256], cs[1][:256], cfg, "fp32")
    t0 = time.perf_counter()
    O.decoder_block(sd, "visual_transformer_blocks.0", x, text, temb, cs[0], cs[1], cfg, "fp32")
    dt = time.perf_counter() - t0
    fl = 2 * n * (6 * 1792**2 + 2 * 1792 * 7168) + 4 * n * n * 1792 + 4 * n * 256 * 1792
    print(f"threads {th:4d}: block N={n}: {dt:.3f} s  {fl / dt / 1e12:.2f} TFLOP/s", flush=True)
