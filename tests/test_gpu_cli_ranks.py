"""The reference's launch line END TO END (README.md:269-276: `python -m torch.distributed.run --nproc-per-node N test.py --config ...`): the factory
`get_T2V_pipeline` (kandinsky/utils.py:23-134) reads LOCAL_RANK / WORLD_SIZE, loads the checkpoints named by a YAML config of the reference's schema
(DiT safetensors, diffusers-style VAE folder, HF text-encoder folders), sets up the multi-process DiT (token shards and / or the CFG pair) and the VAE's
tile distribution, and `Kandinsky5T2VPipeline.__call__` runs text encoding -> sampling -> VAE decode -> uint8 frames on every rank, returning the frames on
rank 0.  Until round 6 nothing executed this path with more than one process.  Here: tiny checkpoints written to a temporary folder (the committed tiny
Qwen2.5-VL / CLIP text encoders + random DiT / VAE weights), P processes on the one GPU over the IPC transport (K5_SP_TRANSPORT=ipc, K5_OVERSUBSCRIBE=1).

  * 2 processes = the CFG pair with one token shard each: the frames must equal the ONE-process run's BIT FOR BIT (each branch's forward is the
    single-handle forward; both handles apply the same combine + Euler update to the same bf16 velocities; VAE tiles are distributed, the blends replicated);
  * 4 processes = CFG pair x 2 token shards: within the tolerance of two summation orders on uint8 frames (mean < 1 grey level, 99 % within 3);
  * the reference CLI itself (kandinsky-5_amd/test.py, same flags as the reference's test.py) under the launcher with 2 processes: exit code 0 and a file.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DIT = dict(in_visual_dim=16, out_visual_dim=16, time_dim=64, patch_size=[1, 2, 2], model_dim=128, ff_dim=256, num_text_blocks=1, num_visual_blocks=2,
           axes_dims=[16, 24, 24], visual_cond=True, in_text_dim=96, in_text_dim2=48)


@pytest.fixture(scope="module")
def tiny_checkpoints(tmp_path_factory):
    import yaml
    from safetensors.torch import save_file
    from kandinsky.models.dit import get_dit
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    from kandinsky.config import Conf
    d = tmp_path_factory.mktemp("tiny_ckpt")
    g = torch.Generator().manual_seed(0)
    dit = get_dit(Conf(DIT))
    sd = {k: (torch.ones_like(v) if k.endswith("norm.weight") else torch.randn(v.shape, generator=g) * 0.05).contiguous() for k, v in dit.state_dict().items()}
    save_file(sd, str(d / "dit.safetensors"))
    vcfg = dict(block_out_channels=[64, 64, 128, 128], norm_num_groups=16)
    vae = AutoencoderKLHunyuanVideo(**vcfg)
    vsd = {}
    for k, p in vae.state_dict().items():
        if "norm" in k and k.endswith("weight"):
            vsd[k] = torch.ones(p.shape)
        elif k.endswith("bias"):
            vsd[k] = torch.zeros(p.shape)
        else:
            vsd[k] = torch.randn(p.shape, generator=g) / (p[0].numel() ** 0.5)
    os.makedirs(d / "vae")
    json.dump(vcfg, open(d / "vae" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in vsd.items()}, str(d / "vae" / "diffusion_pytorch_model.safetensors"))
    meta = json.load(open(os.path.join(GOLD, "text_embed_meta.json")))
    conf = {"model": {"checkpoint_path": str(d / "dit.safetensors"), "num_steps": 3, "guidance_weight": 4.0, "dit_params": DIT,
                      "attention": {"type": "flash", "causal": False, "local": False, "glob": False, "window": 3},
                      "vae": {"name": "hunyuan", "checkpoint_path": str(d)},
                      "text_embedder": {"qwen": {"checkpoint_path": os.path.join(GOLD, "tiny_qwen"), "emb_size": 96, "max_length": meta["max_length_qwen"]},
                                        "clip": {"checkpoint_path": os.path.join(GOLD, "tiny_clip"), "emb_size": 48, "max_length": meta["max_length_clip"]}}},
            "metrics": {"scale_factor": [1.0, 2.0, 2.0], "resolution": 512}, "resolution": 512}
    yaml.safe_dump(conf, open(d / "tiny.yaml", "w"))
    return d


def run(P, script_args, timeout=600):
    env = dict(os.environ, K5_SP_TRANSPORT="ipc", K5_OVERSUBSCRIBE="1", K5_IPC_TIMEOUT_S="120", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if P == 1:
        cmd = [sys.executable] + script_args
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    pr = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        log, _ = pr.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(pr.pid, signal.SIGKILL)
        log, _ = pr.communicate()
        pytest.fail(f"{P} process(es) did not finish within {timeout} s:\n{log[-3000:]}")
    assert pr.returncode == 0, f"exit code {pr.returncode}:\n{log[-4000:]}"
    return log


@pytest.mark.timeout(1500)
def test_factory_and_pipeline_under_the_launcher_equal_the_one_process_run(tiny_checkpoints, tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    worker = os.path.join(ROOT, "tests", "cli_rank_worker.py")
    outs = {}
    for P in (1, 2, 4):
        out = str(tmp_path / f"frames_{P}.pt")
        run(P, [worker, "--config", str(tiny_checkpoints / "tiny.yaml"), "--out", out])
        outs[P] = torch.load(out)
        assert outs[P]["world"] == P and tuple(outs[P]["frames"].shape) == (1, 3, 25, 512, 512)
    assert outs[2]["ipc_ranks"] == 0 and outs[2]["ipc_pair_ranks"] == 2 and outs[2]["ipc_errors"] == 0      # one shard per branch: the pair only
    assert outs[4]["ipc_ranks"] == 2 and outs[4]["ipc_pair_ranks"] == 2 and outs[4]["ipc_errors"] == 0
    one, two, four = (outs[P]["frames"] for P in (1, 2, 4))
    assert torch.equal(two, one), f"2 processes (CFG pair) differ from one process in {(two != one).float().mean().item():.3e} of the pixels"
    diff = (four.int() - one.int()).abs()
    print(f"4 processes (CFG pair x 2 token shards) vs one process, uint8 frames: mean |difference| {diff.float().mean().item():.4f} grey levels, "
          f"within 1: {(diff <= 1).float().mean().item():.4f}, max {int(diff.max())}; the frames are not constant: std {one.float().std().item():.1f}")
    assert one.float().std().item() > 1.0
    assert diff.float().mean().item() < 1.0 and (diff <= 3).float().mean().item() >= 0.99


@pytest.mark.timeout(900)
def test_reference_cli_under_the_launcher(tiny_checkpoints, tmp_path):
    """kandinsky-5_amd/test.py with the reference's flags, 2 processes: rank 0 writes the clip (losslessly as an animated PNG where no mp4 muxer exists)."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    out = tmp_path / "clip.mp4"
    log = run(2, [os.path.join(ROOT, "kandinsky-5_amd", "test.py"), "--config", str(tiny_checkpoints / "tiny.yaml"), "--prompt", "a dog on a red sofa",
                  "--video_duration", "1", "--width", "512", "--height", "512", "--expand_prompt", "0", "--sample_steps", "3", "--output_filename", str(out)])
    assert "Generated video is saved to" in log
    written = [p for p in (out, out.with_suffix(".png")) if p.exists()]
    assert written and written[0].stat().st_size > 10000, (log[-2000:], list(tmp_path.iterdir()))
