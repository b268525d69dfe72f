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

    def __call__(self, text: str, time_length: int = 5, width: int = 768, height: int = 512, seed: int = None,
                 num_steps: int = None, guidance_weight: float = None, scheduler_scale: float = 10.0,
                 negative_caption: str = _NEG, expand_prompts: bool = True, save_path: str = None,
                 progress: bool = True):
        num_steps = self.num_steps if num_steps is None else num_steps
        guidance_weight = self.guidance_weight if guidance_weight is None else guidance_weight
        if seed is None:  # rank 0 draws, everyone agrees (reference :108-118)
            if self.local_dit_rank == 0:
                seed = torch.randint(2 ** 63 - 1, (1,)).to(self.local_dit_rank)
            else:
                seed = torch.empty((1,), dtype=torch.int64).to(self.local_dit_rank)
            if self.world_size > 1:
                torch.distributed.broadcast(seed, 0)
            seed = seed.item()
        if self.resolution != 512:
            raise NotImplementedError("Only 512 resolution is available for now")
        if (height, width) not in self.RESOLUTIONS[self.resolution]:
            raise ValueError(
                f"Wrong height, width pair. Available (height, width) are: {self.RESOLUTIONS[self.resolution]}")

        num_frames = 1 if time_length == 0 else time_length * 24 // 4 + 1
        caption = text
        if expand_prompts:
            if self.local_dit_rank == 0:
                if self.offload:
                    self.text_embedder = self.text_embedder.to(self.device_map["text_embedder"])
                caption = self.expand_prompt(caption)
            if self.world_size > 1:
                caption = [caption]
                torch.distributed.broadcast_object_list(caption, 0)
                caption = caption[0]
        shape = (1, num_frames, height // 8, width // 8, 16)

        images = generate_sample(shape, caption, self.dit, self.vae, self.conf, text_embedder=self.text_embedder,
                                 num_steps=num_steps, guidance_weight=guidance_weight, scheduler_scale=scheduler_scale,
                                 negative_caption=negative_caption, seed=seed, device=self.device_map["dit"],
                                 vae_device=self.device_map["vae"],
                                 text_embedder_device=self.device_map["text_embedder"], progress=progress,
                                 offload=self.offload)
        torch.cuda.empty_cache()

        if self.local_dit_rank != 0:
            return None
        if time_length == 0:
            from .video_io import to_pil_images
            return_images = to_pil_images(images.squeeze(2).cpu())
            if save_path is not None:
                save_path = [save_path] if isinstance(save_path, str) else save_path
                if len(save_path) == len(return_images):
                    for path, image in zip(save_path, return_images):
                        image.save(path)
            return return_images
        if save_path is not None:
            from .video_io import write_video
            save_path = [save_path] if isinstance(save_path, str) else save_path
            if len(save_path) == len(images):
                for path, video in zip(save_path, images):
                    write_video(path, video.permute(1, 2, 3, 0).cpu(), fps=24)
        return images
