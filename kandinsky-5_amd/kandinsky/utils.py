"""Factory — host mirror of kandinsky/utils.py:23-198 (`get_T2V_pipeline`, `get_default_conf`).

Keeps the reference signature, the YAML schema and the safetensors checkpoint layout.  What differs:
no DTensor tensor-parallel wrap (parallelize.py) — multi-GPU is token-sharded sequence parallelism inside
the engine (+ CFG-parallel rank groups, VAE tile distribution); the LOCAL_RANK / WORLD_SIZE launch contract (utils.py:40-45) is kept.
"""
import os
from typing import Union

import torch

from .config import Conf, load_config
from .models.dit import get_dit
from .t2v_pipeline import Kandinsky5T2VPipeline


def get_T2V_pipeline(
    device_map: Union[str, torch.device, dict],
    resolution: int = 512,
    cache_dir: str = "./weights/",
    dit_path: str = None,
    text_encoder_path: str = None,
    text_encoder2_path: str = None,
    vae_path: str = None,
    conf_path: str = None,
    offload: bool = False,
    magcache: bool = False,
) -> Kandinsky5T2VPipeline:
    assert resolution in [512]
    if not isinstance(device_map, dict):
        device_map = {"dit": device_map, "vae": device_map, "text_embedder": device_map}
    try:
        local_rank, world_size = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    except (KeyError, ValueError):
        local_rank, world_size = 0, 1
    assert not (world_size > 1 and offload), "Offloading available only with not parallel inference"
    if world_size > 1:
        for k in ("dit", "vae", "text_embedder"):
            device_map[k] = torch.device(f"cuda:{rank_device_index(local_rank)}")

    os.makedirs(cache_dir, exist_ok=True)
    if conf_path is None:
        # the reference downloads from the HF hub here (utils.py:59-87); offline builds take local paths
        dit_path = dit_path or os.path.join(cache_dir, "model/kandinsky5lite_t2v_sft_5s.safetensors")
        vae_path = vae_path or os.path.join(cache_dir, "vae/")
        text_encoder_path = text_encoder_path or os.path.join(cache_dir, "text_encoder/")
        text_encoder2_path = text_encoder2_path or os.path.join(cache_dir, "text_encoder2/")
        conf = get_default_conf(dit_path, vae_path, text_encoder_path, text_encoder2_path)
    else:
        conf = load_config(conf_path)

    from .models.text_embedders import get_text_embedder
    from .models.vae import build_vae
    from safetensors.torch import load_file

    text_embedder = get_text_embedder(conf.model.text_embedder)
    if not offload:
        text_embedder = text_embedder.to(device=device_map["text_embedder"])
    vae = build_vae(conf.model.vae).eval()
    if not offload:
        vae = vae.to(device=device_map["vae"])

    with torch.device("meta"):
        dit = get_dit(conf.model.dit_params)
    if magcache:  # reference utils.py:107-113
        from .magcache_utils import set_magcache_params
        no_cfg = conf.model.guidance_weight == 1.0
        set_magcache_params(dit, conf.magcache.mag_ratios, conf.model.num_steps, no_cfg)
    state_dict = load_file(conf.model.checkpoint_path)
    dit.load_state_dict(state_dict, assign=True)
    if not offload:
        dit = dit.to(device_map["dit"])

    if world_size > 1:
        # reference utils.py:47-55,121-122 (init_device_mesh + parallelize_dit): one process per GPU, RCCL.  The token axis
        # is sharded instead of the heads; with classifier-free guidance and an even rank count the cond / uncond forwards
        # run on two rank groups (K5_CFG_PARALLEL=0 turns that off); VAE temporal tiles are spread over all ranks.
        import torch.distributed as dist
        from .models.parallelize import parallelize_dit
        if not dist.is_initialized():
            init_rank_process_group(local_rank)
        cfg_parallel = (conf.model.guidance_weight != 1.0 and world_size % 2 == 0
                        and os.environ.get("K5_CFG_PARALLEL", "1") != "0")
        dit = parallelize_dit(dit, local_rank, world_size, device=device_map["dit"], cfg_parallel=cfg_parallel)
        vae.enable_tile_parallel(local_rank, world_size)

    return Kandinsky5T2VPipeline(device_map=device_map, dit=dit, text_embedder=text_embedder, vae=vae,
                                 resolution=resolution, local_dit_rank=local_rank, world_size=world_size, conf=conf,
                                 offload=offload)


def rank_device_index(local_rank: int) -> int:
    """The device of a rank: cuda:LOCAL_RANK (reference utils.py:40-45).  K5_OVERSUBSCRIBE=1 wraps the ranks around the devices
    that exist (P ranks on one GPU: the IPC transport's one-box mode, `bench.py --gpus P --oversubscribe`)."""
    if os.environ.get("K5_OVERSUBSCRIBE", "0") == "1":
        return local_rank % max(torch.cuda.device_count(), 1)
    return local_rank


def init_rank_process_group(local_rank: int):
    """reference utils.py:47-55 (init_device_mesh) under its launch contract.  RCCL transport: backend nccl bound to the rank's
    device.  IPC transport (K5_SP_TRANSPORT=ipc): torch.distributed only carries names and checksums — gloo, which unlike RCCL
    does not mind several ranks on one device."""
    import torch.distributed as dist
    from .models.dit import sp_transport
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if sp_transport() == "ipc":
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{rank_device_index(local_rank)}"))


def get_default_conf(dit_path, vae_path, text_encoder_path, text_encoder2_path) -> Conf:
    """reference utils.py:137-198: the configuration used when no `conf_path` is given = the 5 s SFT model (dense attention,
    50 steps, guidance 5) with the four checkpoint locations filled in."""
    from .config import default_configs
    base = default_configs()["config_5s_sft.yaml"]
    model = dict(base["model"])
    model["checkpoint_path"] = dit_path
    model["vae"] = dict(model["vae"], checkpoint_path=vae_path)
    model["text_embedder"] = {"qwen": dict(model["text_embedder"]["qwen"], checkpoint_path=text_encoder_path),
                              "clip": dict(model["text_embedder"]["clip"], checkpoint_path=text_encoder2_path)}
    return Conf({"model": model, "metrics": {"scale_factor": (1, 2, 2)}, "resolution": 512})
