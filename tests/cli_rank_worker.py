"""One rank (or the only process) of the end-to-end launch-contract test (not a test module: tests/test_gpu_cli_ranks.py runs it plainly and under
`python -m torch.distributed.run --nproc-per-node P`, the reference's launch line README.md:269-276).  It does what the reference's test.py does —
`get_T2V_pipeline(device_map, conf_path=...)` then `pipe(prompt, ...)` (test.py:120-147) — with a FIXED seed, which the CLI has no flag for, so that the
parent test can compare the frames of a one-process run with those of a multi-process run."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    from kandinsky import get_T2V_pipeline
    pipe = get_T2V_pipeline(device_map={"dit": "cuda:0", "vae": "cuda:0", "text_embedder": "cuda:0"}, conf_path=args.config)
    out = pipe("a cat in a blue hat", time_length=1, width=512, height=512, seed=7, num_steps=args.steps, scheduler_scale=5.0, expand_prompts=False,
               save_path=None, progress=False)
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        assert out is not None and out.dtype == torch.uint8
        torch.save({"frames": out.cpu(), "world": int(os.environ.get("WORLD_SIZE", "1")),
                    "ipc_ranks": pipe.dit.get_option("ipc_ranks"), "ipc_pair_ranks": pipe.dit.get_option("ipc_pair_ranks"),
                    "ipc_errors": pipe.dit.get_option("ipc_errors") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else 0}, args.out)
    else:
        assert out is None          # the reference returns the frames on rank 0 only (t2v_pipeline.py:166)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
