"""The reference's public surface end to end: Kandinsky5T2VPipeline.__call__ (t2v_pipeline.py:90-189) ->
generate_sample (generation_utils.py:132-228) -> engine DiT sampling -> engine VAE decode -> uint8 frames.
HF text-encoder weights do not exist offline, so the text embedder is a stub with the reference's `encode` contract
(dict of text_embeds / pooled_embed + cu_seqlens); everything downstream is the product path."""
import pytest
import torch

from kandinsky.config import Conf
from kandinsky.t2v_pipeline import Kandinsky5T2VPipeline

DIT = dict(in_visual_dim=16, out_visual_dim=16, time_dim=64, patch_size=[1, 2, 2], model_dim=128, ff_dim=256,
           num_text_blocks=1, num_visual_blocks=2, axes_dims=[16, 24, 24], visual_cond=True, in_text_dim=96, in_text_dim2=48)


class StubTextEmbedder:
    def __init__(self, device="cpu"):
        self.device = device

    def encode(self, texts, type_of_content="image"):
        g = torch.Generator().manual_seed(len(texts[0]))
        L = 5 + len(texts[0]) % 7
        return ({"text_embeds": torch.randn(L, 96, generator=g), "pooled_embed": torch.randn(1, 48, generator=g)},
                torch.tensor([0, L], dtype=torch.int32))

    def to(self, device):
        return self


def make_conf(attn=None):
    return Conf({"model": {"num_steps": 3, "guidance_weight": 4.0, "dit_params": DIT,
                           "attention": attn or {"type": "flash", "causal": False, "local": False, "glob": False, "window": 3}},
                 "metrics": {"scale_factor": [1.0, 2.0, 2.0]}})


def test_pipeline_argument_errors_match_reference():
    pipe = Kandinsky5T2VPipeline("cpu", dit=None, text_embedder=StubTextEmbedder(), vae=None, conf=make_conf())
    with pytest.raises(ValueError, match="Wrong height, width pair"):
        pipe("a cat", width=640, height=480, seed=1, expand_prompts=False)
    with pytest.raises(ValueError, match="Resolution can be only 512"):
        Kandinsky5T2VPipeline("cpu", dit=None, text_embedder=None, vae=None, resolution=1024, conf=make_conf())
    assert pipe.num_steps == 3 and pipe.guidance_weight == 4.0


@pytest.mark.gpu
@pytest.mark.parametrize("time_length,attn", [(0, None), (1, None),
                                              (1, {"type": "nabla", "P": 0.8, "wT": 3, "wH": 3, "wW": 3, "add_sta": True,
                                                   "method": "topcdf"})])
def test_pipeline_end_to_end_on_engine(time_length, attn, tmp_path):
    from kandinsky.models.dit import get_dit
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    dev = "cuda:0"
    conf = make_conf(attn)
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
    pipe = Kandinsky5T2VPipeline({"dit": dev, "vae": dev, "text_embedder": dev}, dit=dit, text_embedder=StubTextEmbedder(),
                                 vae=vae, conf=conf)
    save = str(tmp_path / ("out.png" if time_length == 0 else "out.mp4"))
    out = pipe("a cat in a blue hat", time_length=time_length, width=512, height=512, seed=7, expand_prompts=False,
               scheduler_scale=5.0, save_path=save)
    if time_length == 0:
        assert isinstance(out, list) and out[0].size == (512, 512)
    else:
        frames = time_length * 24 // 4 + 1
        assert out.dtype == torch.uint8 and tuple(out.shape) == (1, 3, 4 * (frames - 1) + 1, 512, 512)
        assert out.float().std() > 1.0          # not a constant image
    # same seed -> same result (the whole path is deterministic)
    out2 = pipe("a cat in a blue hat", time_length=time_length, width=512, height=512, seed=7, expand_prompts=False,
                scheduler_scale=5.0)
    if time_length:
        assert torch.equal(out, out2)
