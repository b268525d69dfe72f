"""Engine-level parity of the NABLA path AT THE LENGTHS OF BASELINE CONFIGS 4 AND 5 against the REFERENCE: one-block
forward at 93 696 tokens (1464 blocks) and 234 240 tokens (3660 blocks), full width, NABLA P = 0.9 / window (11, 3, 3), through k5_dit_forward —
as ONE handle and as 4 sequence-parallel ranks on one GPU (BASELINE config 4's SP x 4) — compared on the output patches of sampled 64-token
query blocks with the reference's own forward (oracle/gen_golden_nabla_long.py: dit.py:155-181 with its nablaT_v2 map; flex_attention evaluated
exactly on the sampled query blocks against every key block the reference's BlockMask keeps).

Tolerance: rel-L2 <= 3e-2 on the sampled patches — the suite's engine-vs-reference-fp32 bound (bf16 autocast noise), which here also has to
absorb the map's legitimate ambiguity: a bf16 block logit may round the other way on the GPU (as it may on the reference's own GPU matmul), which
moves a block across the cumulative-mass cut for a row; such a block carries ~1 / kept of the row's attention mass."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import k5_oracle as O  # noqa: E402

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def _case(tag):
    from safetensors.torch import load_file
    meta = json.load(open(os.path.join(HERE, "dit_nabla_long_meta.json")))[tag]
    G = load_file(os.path.join(HERE, f"dit_nabla_long_{tag}.safetensors"))
    c = dict(O.LITE_2B, num_visual_blocks=meta.get("visual_blocks", 1), num_text_blocks=1)
    sd = O.synthetic_state_dict(O.DitConfig(**c), seed=meta["weights_seed"])
    for k in sd:
        if k.endswith(("query_norm.weight", "key_norm.weight")):
            sd[k] = torch.full((64,), float(meta["qk_gain"]))
    T, H, W = meta["latent"]
    g = torch.Generator().manual_seed(meta["input_seed"])
    x = torch.randn(T, H, W, 16, generator=g)
    text, pooled = torch.randn(meta["text_len"], 3584, generator=g), torch.randn(1, 768, generator=g)
    pos = [torch.arange(T), torch.arange(H // 2), torch.arange(W // 2)]
    sp = {"P": meta["P"], "wT": meta["window"][0], "wH": meta["window"][1], "wW": meta["window"][2], "to_fractal": True}
    return meta, G, c, sd, (x, text, pooled, pos, sp)


def _patches(out, meta, blocks):
    T, H, W = meta["latent"]
    Hb, Wb = H // 16, W // 16
    res = []
    for b in blocks.tolist():
        t, hb, wb = b // (Hb * Wb), (b // Wb) % Hb, b % Wb
        res.append(out[t, 16 * hb:16 * hb + 16, 16 * wb:16 * wb + 16, :].float().cpu())
    return torch.stack(res)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("tag", ["c4", "c5", "c4d"])
def test_nabla_forward_at_config_length_vs_reference_golden(tag):
    """c4 / c5: one visual block.  c4d (round 5): THREE visual blocks at config 4's length, the reference's attention evaluated on every row, so
    the second and third blocks' maps are built from — and their attention runs over — activations that already went through a NABLA block."""
    from kandinsky.models.dit import DiffusionTransformer3D
    meta, G, c, sd, (x, text, pooled, pos, sp) = _case(tag)
    dit = DiffusionTransformer3D(**c)
    dit.load_state_dict(sd, assign=True)
    dit = dit.to("cuda:0")
    L = meta["text_len"]
    out = dit(x.cuda(), text.cuda(), pooled.cuda(), torch.tensor([meta["time"]]), pos, torch.arange(L), scale_factor=(1.0, 2.0, 2.0), sparse_params=sp)
    torch.cuda.synchronize()
    n_fixed, n_online = dit.attn_variant_counts()
    got = _patches(out, meta, G["sampled_blocks"])
    r = rel(got, G["patches"])
    worst = max(rel(got[i], G["patches"][i]) for i in range(got.shape[0]))
    print(f"NABLA at BASELINE config {tag[1]}'s length, {meta.get('visual_blocks', 1)} visual block(s) (N = {meta['tokens']}, {meta['blocks']} blocks, reference map density {meta['kept_density_on_sampled_rows']:.3f} on "
          f"the sampled rows): engine vs reference fp32 {r:.3e} over {got.shape[0]} sampled blocks (worst block {worst:.3e}); heads fixed / online {n_fixed} / {n_online}")
    assert torch.isfinite(out.float()).all()
    assert r <= 3e-2, r
    assert worst <= 6e-2, worst
    dit._destroy_engine(force=True)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("tag", ["c4", "c4d"])
def test_nabla_config4_as_4_ranks_vs_reference_golden(tag):
    """BASELINE config 4 as it is deployed — sequence-parallel x 4 — on one GPU through the loopback group: every rank's gathered velocity
    against the reference golden on the sampled blocks (blocks of every rank's token shard are among them)."""
    from test_gpu_loopback import run_ranks
    from kandinsky.models.dit import DiffusionTransformer3D
    meta, G, c, sd, (x, text, pooled, pos, sp) = _case(tag)
    L = meta["text_len"]

    def make():
        d = DiffusionTransformer3D(**c)
        d.load_state_dict(sd, assign=True)
        return d.to("cuda:0")

    def call(d, r):
        return d(x.cuda(), text.cuda(), pooled.cuda(), torch.tensor([meta["time"]]), pos, torch.arange(L), scale_factor=(1.0, 2.0, 2.0), sparse_params=sp)

    outs = run_ranks(4, make, call)
    for r_ in range(1, 4):
        assert torch.equal(outs[r_], outs[0]), f"rank {r_} differs from rank 0"
    got = _patches(outs[0], meta, G["sampled_blocks"])
    r = rel(got, G["patches"])
    shard = meta["blocks"] // 4
    print(f"config 4 as 4 ranks, {meta.get('visual_blocks', 1)} visual block(s): vs reference fp32 {r:.3e}; sampled blocks per rank shard: {[int(((G['sampled_blocks'] // shard).clamp(max=3) == k).sum()) for k in range(4)]}")
    assert r <= 3e-2, r
