"""Text-embedding handoff with REAL HF model classes: tiny random-init Qwen2.5-VL / CLIP-text checkpoints committed
under tests/golden/tiny_qwen, tiny_clip (data: config + weights + byte-level tokenizer files).  The expected outputs were produced by
the reference's own Qwen2_5_VLTextEmbedder.__call__ / ClipTextEmbedder.__call__ / Kandinsky5TextEmbedder.encode
(reference text_embedders.py:19-31,67-107) on those checkpoints — oracle/gen_golden_text.py — and the host mirror, built through its
real from_pretrained constructors, must reproduce them: prompt template, crop offset, truncation, ragged batch, cu_seqlens."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from kandinsky.config import Conf
from kandinsky.models.text_embedders import get_text_embedder

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def embedder():
    meta = json.load(open(os.path.join(GOLD, "text_embed_meta.json")))
    conf = Conf({"qwen": {"checkpoint_path": os.path.join(GOLD, "tiny_qwen"), "max_length": meta["max_length_qwen"]},
                 "clip": {"checkpoint_path": os.path.join(GOLD, "tiny_clip"), "max_length": meta["max_length_clip"]}})
    return get_text_embedder(conf, "cpu"), meta, load_file(os.path.join(GOLD, "text_embed_tiny.safetensors"))


def test_encode_matches_reference_outputs(embedder):
    emb, meta, gold = embedder
    assert len(meta["cases"]) == 6
    for case in meta["cases"]:
        enc, cu = emb.encode(case["texts"], type_of_content=case["type_of_content"])
        tag = case["tag"]
        assert cu.dtype == torch.int32 and cu.tolist() == case["cu_seqlens"] == gold[tag + "_cu_seqlens"].tolist()
        te = enc["text_embeds"]
        assert te.dtype == torch.bfloat16 and tuple(te.shape) == tuple(gold[tag + "_text_embeds"].shape)
        # same weights, same tokens, same sdpa CPU kernels: bf16 results differ at most by blocking-dependent rounding
        assert (te.float() - gold[tag + "_text_embeds"].float()).abs().max().item() <= 6e-2
        assert (te.float() - gold[tag + "_text_embeds"].float()).abs().mean().item() <= 2e-3
        pe = enc["pooled_embed"]
        assert tuple(pe.shape) == (len(case["texts"]), 48)
        assert torch.allclose(pe.float(), gold[tag + "_pooled_embed"], atol=1e-4, rtol=1e-4)


def test_crop_and_truncation_contract(embedder):
    emb, meta, _ = embedder
    q = emb.embedder
    crop_v, crop_i = q.PROMPT_TEMPLATE["crop_start"]["video"], q.PROMPT_TEMPLATE["crop_start"]["image"]
    assert (crop_v, crop_i) == (129, 41)                                  # reference text_embedders.py:53
    # a prompt longer than max_length: exactly max_length embeddings survive (max_length + crop tokens kept, crop dropped)
    _, cu = q(["y" * 3000], type_of_content="video")
    assert cu.tolist() == [0, meta["max_length_qwen"]]
    # ragged batch: right padding is dropped by the attention mask, one row of cu_seqlens per prompt
    e, cu = q(["short", "a somewhat longer prompt"], type_of_content="image")
    assert cu[1] - cu[0] < cu[2] - cu[1] and e.shape[0] == int(cu[-1])
    assert cu[2] - cu[1] - (cu[1] - cu[0]) == len("a somewhat longer prompt") - len("short")   # one token per byte in the tiny vocabulary


def test_text_only_processor_refuses_pixels(embedder):
    emb = embedder[0]
    proc = emb.embedder.processor
    if type(proc).__name__ != "_TextOnlyProcessor":
        pytest.skip("the full VL processor could be built here")
    with pytest.raises(ValueError, match="text-only"):
        proc(text=["a"], images=[object()])


@pytest.mark.gpu
def test_pipeline_with_real_embedder_classes(embedder, tmp_path):
    """Kandinsky5T2VPipeline end to end with the HF-backed embedder instead of a stub: (L, 96) bf16 token embeddings + (1, 48)
    pooled go through generate_sample into the engine (in_text_dim = 96, in_text_dim2 = 48 are the tiny checkpoints' widths)."""
    from kandinsky.models.dit import get_dit
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    from kandinsky.t2v_pipeline import Kandinsky5T2VPipeline
    from test_pipeline import DIT, make_conf
    emb = embedder[0]
    dev = "cuda:0"
    conf = make_conf()
    dit = get_dit(conf.model.dit_params)
    g = torch.Generator().manual_seed(0)
    sd = {k: (torch.ones_like(v) if k.endswith("norm.weight") else torch.randn(v.shape, generator=g) * 0.05)
          for k, v in dit.state_dict().items()}
    dit.load_state_dict(sd, assign=True)
    dit = dit.to(dev)
    vae = AutoencoderKLHunyuanVideo(block_out_channels=(64, 64, 128, 128), norm_num_groups=16)
    vsd = {}
    for k, p in vae.state_dict().items():
        if "norm" in k and k.endswith("weight"):
            vsd[k] = torch.ones(p.shape)
        elif k.endswith("bias"):
            vsd[k] = torch.zeros(p.shape)
        else:
            vsd[k] = torch.randn(p.shape, generator=g) / (p[0].numel() ** 0.5)
    vae.load_state_dict(vsd, assign=True)
    vae = vae.eval().to(dev)
    assert DIT["in_text_dim"] == 96 and DIT["in_text_dim2"] == 48
    pipe = Kandinsky5T2VPipeline({"dit": dev, "vae": dev, "text_embedder": "cpu"}, dit=dit, text_embedder=emb, vae=vae, conf=conf)
    outs = []
    for prompt in ("a cat in a blue hat", "a dog on a red sofa"):
        out = pipe(prompt, time_length=1, width=512, height=512, seed=7, expand_prompts=False, scheduler_scale=5.0,
                   save_path=str(tmp_path / "o.mp4"))
        assert out.dtype == torch.uint8 and tuple(out.shape) == (1, 3, 25, 512, 512)
        outs.append(out)
    assert (outs[0] != outs[1]).float().mean().item() > 0.01     # the prompt reaches the sampler: different text, different clip
