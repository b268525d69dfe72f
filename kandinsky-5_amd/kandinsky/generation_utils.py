"""Sampler — host mirror of kandinsky/generation_utils.py (same function names and signatures).

`generate` keeps the reference's positional signature (generation_utils.py:81-96) and its noise /
sigma-schedule construction, but runs the Euler / CFG loop on the MI355X engine: in one C call
(`DiffusionTransformer3D.sample` -> k5_sample) when `model` is the engine-backed DiT, otherwise step by
step through `model(...)` (duck-typed models, e.g. a MagCache wrapper) with the fused CFG+Euler kernel.
"""
import os

os.environ.setdefault("TOKENIZERS_PARALLELISM", "False")

import torch

from . import _engine as E


def _attr(obj, name, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def get_sparse_params(conf, batch_embeds, device):
    """reference generation_utils.py:10-36.  The STA mask itself is rebuilt on device by the engine from
    (wT, wH, wW); the dict keeps the reference's keys so callers can introspect it."""
    patch = conf.model.dit_params.patch_size
    assert patch[0] == 1
    T, H, W, _ = batch_embeds["visual"].shape
    T, H, W = T // patch[0], H // patch[1], W // patch[2]
    attn = conf.model.attention
    if _attr(attn, "type") == "nabla":
        if H % 8 or W % 8:
            raise ValueError("nabla attention needs latent height/width divisible by 16 (8x8 token tiles)")
        return {
            "sta_mask": None,  # built inside the engine
            "attention_type": _attr(attn, "type"),
            "to_fractal": True,
            "P": _attr(attn, "P"),
            "wT": _attr(attn, "wT"),
            "wW": _attr(attn, "wW"),
            "wH": _attr(attn, "wH"),
            "add_sta": _attr(attn, "add_sta"),
            "visual_shape": (T, H, W),
            "method": _attr(attn, "method", "topcdf"),
        }
    return None


@torch.no_grad()
def get_velocity(dit, x, t, text_embeds, null_text_embeds, visual_rope_pos, text_rope_pos, null_text_rope_pos,
                 guidance_weight, conf, sparse_params=None):
    """reference generation_utils.py:39-77 (two forwards + bf16 CFG combine)."""
    pred_velocity = dit(x, text_embeds["text_embeds"], text_embeds["pooled_embed"], t * 1000, visual_rope_pos,
                        text_rope_pos, scale_factor=conf.metrics.scale_factor, sparse_params=sparse_params)
    if abs(guidance_weight - 1.0) > 1e-6:
        uncond_pred_velocity = dit(x, null_text_embeds["text_embeds"], null_text_embeds["pooled_embed"], t * 1000,
                                   visual_rope_pos, null_text_rope_pos, scale_factor=conf.metrics.scale_factor,
                                   sparse_params=sparse_params)
        pred_velocity = uncond_pred_velocity + guidance_weight * (pred_velocity - uncond_pred_velocity)
    return pred_velocity


def sigma_schedule(num_steps, scheduler_scale, device="cpu"):
    """reference generation_utils.py:102-103"""
    timesteps = torch.linspace(1, 0, num_steps + 1, device=device)
    return scheduler_scale * timesteps / (1 + (scheduler_scale - 1) * timesteps)


@torch.no_grad()
def generate(model, device, shape, num_steps, text_embeds, null_text_embeds, visual_rope_pos, text_rope_pos,
             null_text_rope_pos, guidance_weight, scheduler_scale, conf, progress=False, seed=6554, noise=None):
    """reference generation_utils.py:80-129.  `noise` (optional, extension) overrides the seeded draw."""
    if noise is None:
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        img = torch.randn(*shape, device=device, generator=g)
    else:
        img = noise.to(device=device, dtype=torch.float32).clone()
    img = img.contiguous()

    sparse_params = get_sparse_params(conf, {"visual": img}, device)
    timesteps = sigma_schedule(num_steps, scheduler_scale, device=device).cpu()  # one sync, before the loop

    from .models.dit import DiffusionTransformer3D
    if isinstance(model, torch.nn.Module):      # per-step paths below: a new sampling run starts with no softmax-form memory (k5_sample resets its own)
        for m in model.modules():
            if isinstance(m, DiffusionTransformer3D):
                m.reset_softmax_memory()
    cfg_on = abs(guidance_weight - 1.0) > 1e-6
    cfg_parallel = getattr(model, "_cfg_parallel", None)
    if cfg_parallel is not None and cfg_on and getattr(model, "_cfg_pair", None) is None:
        # CFG-parallel for a model WITHOUT the engine-side pair (a wrapped / duck-typed model): this rank's group
        # runs ONE of the two forwards; the pair exchanges the velocities (6 MB at 5 s) over torch.distributed and every rank applies
        # the identical bf16 combine + Euler update.  A DiffusionTransformer3D set up by parallelize_dit does the same INSIDE
        # k5_sample (k5_dit_cfg_pair_init) and takes the fused path below.
        from .models.parallelize import exchange_velocity
        branch, pair_group = cfg_parallel
        mine, mine_pos = (text_embeds, text_rope_pos) if branch == 0 else (null_text_embeds, null_text_rope_pos)
        both = None
        for timestep, timestep_diff in zip(timesteps[:-1].tolist(), torch.diff(timesteps).tolist()):
            v = model(img, mine["text_embeds"], mine["pooled_embed"], torch.tensor([timestep]) * 1000, visual_rope_pos,
                      mine_pos, scale_factor=conf.metrics.scale_factor, sparse_params=sparse_params)
            if both is None:
                both = torch.empty((2,) + tuple(v.shape), dtype=v.dtype, device=v.device)
            vc, vu = exchange_velocity(v, pair_group, out=both)
            E.cfg_euler_(img, vc, vu, guidance_weight, timestep_diff)
        return img
    if type(model) is DiffusionTransformer3D and model.visual_cond in (True, False):
        # whole loop inside the engine: no per-step host work at all
        model.sample(img, timesteps.tolist(), text_embeds, null_text_embeds, visual_rope_pos, text_rope_pos,
                     null_text_rope_pos, guidance_weight, scale_factor=conf.metrics.scale_factor,
                     sparse_params=sparse_params)
        return img

    for timestep, timestep_diff in zip(timesteps[:-1].tolist(), torch.diff(timesteps).tolist()):
        if model.visual_cond:
            visual_cond = torch.zeros_like(img)
            visual_cond_mask = torch.zeros([*img.shape[:-1], 1], dtype=img.dtype, device=img.device)
            model_input = torch.cat([img, visual_cond, visual_cond_mask], dim=-1)
        else:
            model_input = img
        t1000 = torch.tensor([timestep]) * 1000
        v = model(model_input, text_embeds["text_embeds"], text_embeds["pooled_embed"], t1000, visual_rope_pos,
                  text_rope_pos, scale_factor=conf.metrics.scale_factor, sparse_params=sparse_params)
        u = None
        if cfg_on:
            u = model(model_input, null_text_embeds["text_embeds"], null_text_embeds["pooled_embed"], t1000,
                      visual_rope_pos, null_text_rope_pos, scale_factor=conf.metrics.scale_factor,
                      sparse_params=sparse_params)
        E.cfg_euler_(img, v.contiguous(), None if u is None else u.contiguous(), guidance_weight, timestep_diff)
    return img


def _encode_prompts(text_embedder, prompts, kind, device):
    """[(embeds dict on `device`, number of text tokens)] for each prompt (reference generation_utils.py:153-176)."""
    out = []
    with torch.no_grad():
        for prompt in prompts:
            embeds, cu = text_embedder.encode([prompt], type_of_content=kind)
            out.append(({name: t.to(device=device) for name, t in embeds.items()}, int(cu[-1])))
    return out


def latent_to_uint8(latent, vae, batch, vae_device):
    """Latent (batch*T, H, W, C) -> uint8 frames (batch, 3, F, 8H, 8W): un-scale, channels first, VAE decode, clamp to [-1, 1],
    map to 0..255 (reference generation_utils.py:209-222)."""
    T = latent.shape[0] // batch
    z = latent.reshape(batch, T, *latent.shape[1:]).to(device=vae_device)
    z = (z / vae.config.scaling_factor).permute(0, 4, 1, 2, 3)
    frames = vae.decode(z).sample
    return frames_to_uint8(frames)


def frames_to_uint8(frames):
    """((frames.clamp(-1, 1) + 1) * 127.5).to(torch.uint8) (reference generation_utils.py:222-224).  bf16 frames on the GPU: one pass of the engine's
    kernel with torch's rounding after every elementwise op (k5_frames_to_uint8) instead of four over the whole video; anything else: torch."""
    if frames.is_cuda and frames.dtype == torch.bfloat16 and frames.numel() % 8 == 0:
        from . import _engine as E
        x = frames if frames.is_contiguous() else frames.contiguous()
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            E.check(E.lib().k5_frames_to_uint8(x.data_ptr(), out.data_ptr(), x.numel(), E.stream_ptr(x.device)), "k5_frames_to_uint8")
        return out
    return ((frames.clamp(-1.0, 1.0) + 1.0) * 127.5).to(torch.uint8)


def generate_sample(shape, caption, dit, vae, conf, text_embedder, num_steps=25, guidance_weight=5.0,
                    scheduler_scale=1, negative_caption="", seed=6554, device="cuda", vae_device="cuda",
                    text_embedder_device="cuda", progress=True, offload=False):
    """reference generation_utils.py:132-228 (same signature): text encode -> generate -> VAE decode -> uint8.
    With `offload` each of the three models visits the GPU only for its own stage."""
    batch, frames, height, width, channels = shape
    kind = "image" if frames == 1 else "video"
    (cond, n_cond), (uncond, n_uncond) = _encode_prompts(text_embedder, (caption, negative_caption), kind, device)
    if offload:
        text_embedder = text_embedder.to("cpu")

    patch = conf.model.dit_params.patch_size
    grid = [torch.arange(frames), torch.arange(height // patch[1]), torch.arange(width // patch[2])]
    if offload:
        dit.to(device, non_blocking=True)
    with torch.no_grad():
        latent = generate(dit, device, (batch * frames, height, width, channels), num_steps, cond, uncond, grid,
                          torch.arange(n_cond), torch.arange(n_uncond), guidance_weight, scheduler_scale, conf, seed=seed,
                          progress=progress)
    if offload:
        dit.to("cpu", non_blocking=True)
        torch.cuda.empty_cache()
        vae.to(vae_device, non_blocking=True)
    with torch.no_grad():
        images = latent_to_uint8(latent, vae, batch, vae_device)
    if offload:
        vae.to("cpu", non_blocking=True)
        torch.cuda.empty_cache()
    return images
