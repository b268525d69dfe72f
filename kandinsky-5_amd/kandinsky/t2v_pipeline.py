"""Kandinsky5T2VPipeline — host mirror of kandinsky/t2v_pipeline.py:10-189 (same ctor, same __call__)."""
from typing import Union

import torch

from .generation_utils import generate_sample

_NEG = ("Static, 2D cartoon, cartoon, 2d animation, paintings, images, worst quality, low quality, ugly, "
        "deformed, walking backwards")


class Kandinsky5T2VPipeline:
    def __init__(self, device_map: Union[str, torch.device, dict], dit, text_embedder, vae, resolution: int = 512,
                 local_dit_rank: int = 0, world_size: int = 1, conf=None, offload: bool = False):
        if resolution not in [512]:
            raise ValueError("Resolution can be only 512")
        self.dit, self.text_embedder, self.vae = dit, text_embedder, vae
        self.resolution = resolution
        self.device_map = device_map
        self.local_dit_rank, self.world_size = local_dit_rank, world_size
        self.conf = conf
        self.num_steps = conf.model.num_steps
        self.guidance_weight = conf.model.guidance_weight
        self.offload = offload
        self.RESOLUTIONS = {512: [(512, 512), (512, 768), (768, 512)]}

    def expand_prompt(self, prompt):
        """Prompt beautification through the Qwen2.5-VL chat model (reference t2v_pipeline.py:47-88).
        Delegated to the text embedder wrapper, which owns the HF model."""
        return self.text_embedder.expand_prompt(prompt)

    # ------------------------------------------------------------------ helpers of __call__
    def _agree_on(self, value_fn, as_object=False):
        """rank 0 computes `value_fn()`, every rank returns the same value (reference t2v_pipeline.py:108-118,131-141)."""
        mine = value_fn() if self.local_dit_rank == 0 else None
        if self.world_size <= 1:
            return mine
        if as_object:
            box = [mine]
            torch.distributed.broadcast_object_list(box, 0)
            return box[0]
        # the reference moves the scalar to cuda:LOCAL_RANK (t2v_pipeline.py:112); here the rank's device is whatever the factory assigned it
        # (K5_OVERSUBSCRIBE wraps ranks around the devices that exist) and a host-side process group (gloo: the IPC transport) takes it on the CPU
        host = torch.distributed.get_backend() == "gloo"
        t = torch.tensor([mine], dtype=torch.int64) if mine is not None else torch.empty(1, dtype=torch.int64)
        if not host:
            t = t.to(torch.device(self.device_map["dit"]))
        torch.distributed.broadcast(t, 0)
        return int(t.item())

    def _check_size(self, height, width):
        if self.resolution != 512:
            raise NotImplementedError("Only 512 resolution is available for now")
        allowed = self.RESOLUTIONS[self.resolution]
        if (height, width) not in allowed:
            raise ValueError(f"Wrong height, width pair. Available (height, width) are: {allowed}")

    def _beautified(self, prompt):
        if self.offload:
            self.text_embedder = self.text_embedder.to(self.device_map["text_embedder"])
        return self.expand_prompt(prompt)

    @staticmethod
    def _save(images, time_length, save_path):
        """uint8 (B,3,F,H,W) -> PIL list (image mode) or the tensor itself; optional PNG / video files (reference :166-189)."""
        paths = None if save_path is None else ([save_path] if isinstance(save_path, str) else list(save_path))
        if time_length == 0:
            from .video_io import to_pil_images
            pics = to_pil_images(images.squeeze(2).cpu())
            if paths is not None and len(paths) == len(pics):
                for path, pic in zip(paths, pics):
                    pic.save(path)
            return pics
        if paths is not None and len(paths) == len(images):
            from .video_io import write_video
            for path, clip in zip(paths, images):
                write_video(path, clip.permute(1, 2, 3, 0).cpu(), fps=24)
        return images

    def __call__(self, text: str, time_length: int = 5, width: int = 768, height: int = 512, seed: int = None,
                 num_steps: int = None, guidance_weight: float = None, scheduler_scale: float = 10.0,
                 negative_caption: str = _NEG, expand_prompts: bool = True, save_path: str = None,
                 progress: bool = True):
        """reference t2v_pipeline.py:90-189 (same arguments, defaults, errors and return values: uint8 tensor (1,3,F,H,W) on
        rank 0 / list of PIL images for time_length = 0, None on the other ranks)."""
        steps = self.num_steps if num_steps is None else num_steps
        weight = self.guidance_weight if guidance_weight is None else guidance_weight
        if seed is None:
            seed = self._agree_on(lambda: int(torch.randint(2 ** 63 - 1, (1,)).item()))
        self._check_size(height, width)
        frames = 1 if time_length == 0 else time_length * 24 // 4 + 1
        caption = self._agree_on(lambda: self._beautified(text), as_object=True) if expand_prompts else text

        images = generate_sample((1, frames, height // 8, width // 8, 16), caption, self.dit, self.vae, self.conf,
                                 text_embedder=self.text_embedder, num_steps=steps, guidance_weight=weight,
                                 scheduler_scale=scheduler_scale, negative_caption=negative_caption, seed=seed,
                                 device=self.device_map["dit"], vae_device=self.device_map["vae"],
                                 text_embedder_device=self.device_map["text_embedder"], progress=progress, offload=self.offload)
        torch.cuda.empty_cache()
        return self._save(images, time_length, save_path) if self.local_dit_rank == 0 else None
