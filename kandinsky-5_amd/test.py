"""Command-line entry point — same flags and defaults as the reference's test.py (reference test.py:32-147) so that the
README launch lines keep working against this package:

    python test.py --prompt "a cat in a blue hat" --config ./configs/config_5s_sft.yaml
    PYTHONPATH=. torchrun --nproc-per-node 8 --master-addr 127.0.0.1 test.py --config ./configs/config_10s_sft.yaml ...

Multi-GPU: one process per GPU (LOCAL_RANK / WORLD_SIZE from the launcher); `get_T2V_pipeline` sets up token-sharded
sequence parallelism (+ CFG-parallel rank groups, VAE tile distribution) over RCCL.
"""
import argparse
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SUPPORTED_SIZES = [(512, 512), (512, 768), (768, 512)]
NEGATIVE = ("Static, 2D cartoon, cartoon, 2d animation, paintings, images, worst quality, low quality, ugly, deformed, "
            "walking backwards")


def build_parser():
    p = argparse.ArgumentParser(description="Generate a video with Kandinsky 5 (MI355X engine)")
    p.add_argument("--local-rank", type=int, help="set by the launcher (one process per GPU)")
    p.add_argument("--config", type=str, default="./configs/config_5s_sft.yaml", help="YAML config (reference schema); a missing default name is created")
    p.add_argument("--prompt", type=str, default="a cat in a blue hat", help="text prompt")
    p.add_argument("--negative_prompt", type=str, default=NEGATIVE, help="negative prompt of the unconditional branch")
    p.add_argument("--width", type=int, default=768, choices=[768, 512], help="frame width in pixels")
    p.add_argument("--height", type=int, default=512, choices=[768, 512], help="frame height in pixels")
    p.add_argument("--video_duration", type=int, default=5, help="clip length in seconds (0 = a single image)")
    p.add_argument("--expand_prompt", type=int, default=1, help="1 = rewrite the prompt with the Qwen2.5-VL chat model first")
    p.add_argument("--sample_steps", type=int, default=None, help="number of Euler steps (default: the config's)")
    p.add_argument("--guidance_weight", type=float, default=None, help="classifier-free guidance weight (default: the config's)")
    p.add_argument("--scheduler_scale", type=float, default=5.0, help="sigma-schedule scale s in s*t/(1+(s-1)*t)")
    p.add_argument("--output_filename", type=str, default="./test.mp4", help="output path (.mp4 / .avi / .png)")
    p.add_argument("--offload", action="store_true", default=False, help="keep only the active model on the GPU")
    p.add_argument("--magcache", action="store_true", default=False, help="MagCache: skip the visual blocks on low-error steps (50-step configs)")
    return p


def validate_args(args):
    if (args.width, args.height) not in SUPPORTED_SIZES:
        raise NotImplementedError(f"Provided size of video is not supported: {(args.width, args.height)}")


def main(argv=None):
    warnings.filterwarnings("ignore")
    args = build_parser().parse_args(argv)
    validate_args(args)
    from kandinsky import get_T2V_pipeline
    pipe = get_T2V_pipeline(device_map={"dit": "cuda:0", "vae": "cuda:0", "text_embedder": "cuda:0"}, conf_path=args.config,
                            offload=args.offload, magcache=args.magcache)
    if args.output_filename is None:
        args.output_filename = "./" + args.prompt.replace(" ", "_") + ".mp4"
    t0 = time.perf_counter()
    pipe(args.prompt, time_length=args.video_duration, width=args.width, height=args.height, num_steps=args.sample_steps,
         guidance_weight=args.guidance_weight, scheduler_scale=args.scheduler_scale, expand_prompts=args.expand_prompt,
         negative_caption=args.negative_prompt, save_path=args.output_filename)
    print(f"TIME ELAPSED: {time.perf_counter() - t0}")
    print(f"Generated video is saved to {args.output_filename}")


if __name__ == "__main__":
    main()
