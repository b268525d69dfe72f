"""One RANK of the cross-process run of the sharded path (not a test module: tests/test_gpu_ipc_ranks.py launches it as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 --master-port <free> tests/ipc_rank_worker.py ...`,
the reference's launch line, README.md:269-276).

What runs here is the launch contract of the reference end to end (kandinsky/utils.py:40-55): LOCAL_RANK / WORLD_SIZE from the launcher ->
the rank's device -> torch.distributed process group -> parallelize_dit(model, ...) (kandinsky/models/parallelize.py:11-102, replaced by token
shards) -> the engine's communicator -> sample.  Transport: the engine's IPC group (K5_SP_TRANSPORT=ipc), all ranks on the devices that exist
(K5_OVERSUBSCRIBE=1: on a one-GPU box every rank is a separate PROCESS on device 0 — RCCL cannot do that).

Workload = BASELINE config 1 in full (tests/golden/dit_fulldepth_meta.json c1 / n1: latent (13, 32, 32), 32 visual blocks, NFE 16), the same
weights / noise / prompt streams as tests/test_gpu_fulldepth.py, so that the parent test can compare the latent with the reference's generate()
goldens and, bit for bit, with loopback ranks of the same size.  Each rank writes its final latent; rank 0 also writes the rank_check."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
HERE = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, choices=("c1", "n1w1", "n1w5", "c5", "fwd_c4d", "fwd_c5"))
    ap.add_argument("--graph", action="store_true", help="k5_dit_set_graph: one captured step replayed (the IPC collectives inside the capture)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--slices", type=int, default=1)
    ap.add_argument("--cfg-parallel", action="store_true")
    ap.add_argument("--ulysses", action="store_true", help="engine option sp_mode = 1: the all-to-all schedule (heads % ranks == 0)")
    ap.add_argument("--tiny", action="store_true", help="a 2-block model instead of the 2B one (quick plumbing check)")
    ap.add_argument("--die-before-sample", type=int, default=-1, metavar="RANK", help="failure drill: this rank leaves after the communicator is up")
    args = ap.parse_args()
    os.environ.setdefault("K5_SP_TRANSPORT", "ipc")
    os.environ.setdefault("K5_OVERSUBSCRIBE", "1")
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(16)   # P ranks regenerate 2 B weights from the seed side by side on the host cores

    import torch.distributed as dist
    from oracle import k5_oracle as O
    from kandinsky.generation_utils import sigma_schedule
    from kandinsky.models.dit import DiffusionTransformer3D
    from kandinsky.models.parallelize import parallelize_dit
    from kandinsky.utils import init_rank_process_group, rank_device_index

    dev = torch.device("cuda", rank_device_index(local_rank))
    torch.cuda.set_device(dev)
    init_rank_process_group(local_rank)
    if args.case.startswith("fwd_"):
        # ONE forward at the length of BASELINE config 4 / 5 (93 696 / 234 240 tokens, NABLA) against the reference golden of tests/test_gpu_nabla_long.py:
        # config 4 is "sequence-parallel over 4 x MI355X" by name — here its 4 ranks are 4 processes
        import importlib.util
        spec = importlib.util.spec_from_file_location("k5_nabla_long", os.path.join(ROOT, "tests", "test_gpu_nabla_long.py"))
        nl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(nl)
        tag = args.case[4:]
        m, G, c, sd, (x, text, pooled, pos, sp) = nl._case(tag)
        dit = DiffusionTransformer3D(**c)
        dit.load_state_dict(sd, assign=True)
        dit.engine(dev)
        del sd
        parallelize_dit(dit, rank, world, device=dev, cfg_parallel=False)
        assert dit.get_option("ipc_ranks") == world
        dist.barrier()
        out = dit(x.to(dev), text.to(dev), pooled.to(dev), torch.tensor([m["time"]]), pos, torch.arange(m["text_len"]), scale_factor=(1.0, 2.0, 2.0), sparse_params=sp)
        torch.cuda.synchronize(dev)
        errs = dit.get_option("ipc_errors")
        os.makedirs(args.out, exist_ok=True)
        torch.save(nl._patches(out, m, G["sampled_blocks"]), os.path.join(args.out, f"patches_rank{rank}.pt"))
        cs = torch.stack([out.double().sum(), out.double().abs().sum(), out.view(torch.int16).to(torch.int64).sum().double()]).cpu()
        allcs = [torch.empty_like(cs) for _ in range(world)]
        dist.all_gather(allcs, cs)
        if rank == 0:
            json.dump({"identical": all(torch.equal(allcs[0], v) for v in allcs), "ipc_errors": errs, "collectives": dit.get_option("ipc_collectives"),
                       "finite": bool(torch.isfinite(out.float()).all())}, open(os.path.join(args.out, "fwd_check.json"), "w"))
        dist.barrier()
        dit._destroy_engine(force=True)
        dist.destroy_process_group()
        if errs:
            raise SystemExit(f"rank {rank}: an IPC flag wait timed out")
        return
    meta = json.load(open(os.path.join(HERE, "dit_fulldepth_meta.json")))
    if args.case == "c5":
        # BASELINE config 5's shape as ONE configuration (tests/test_gpu_loopback.py::test_config5_cfg_parallel_2x4_on_one_gpu): 1280x768 10 s latent
        # (61, 96, 160) = 234 240 tokens = 3660 blocks of 64, NABLA P 0.9 window (11, 3, 3), guidance 5, full width, ONE visual block, 4 Euler steps
        c = {"latent": [61, 96, 160], "L": 48, "Lnull": 8, "steps": 4, "s": 10.0, "seed": 13, "xseed": 14, "P": 0.9, "win": [11, 3, 3]}
        w, sparse = 5.0, {"P": 0.9, "wT": 11, "wH": 3, "wW": 3, "to_fractal": True}
        meta = dict(meta, weights_seed=4, qk_gain=1.0)
    elif args.case == "c1":
        c, w, sparse = meta["c1"], meta["c1"]["w"], None
    else:
        c = meta["n1"]
        w = 1.0 if args.case == "n1w1" else 5.0
        sparse = {"P": c["P"], "wT": c["win"][0], "wH": c["win"][1], "wW": c["win"][2], "to_fractal": True}
    cfg = dict(O.LITE_2B)
    if args.tiny:
        cfg.update(num_visual_blocks=2, num_text_blocks=1)
    if args.case == "c5":
        cfg.update(num_visual_blocks=1, num_text_blocks=1)
    sd = O.synthetic_state_dict(O.DitConfig(**cfg), seed=meta["weights_seed"])
    for k in sd:
        if k.endswith(("query_norm.weight", "key_norm.weight")):
            sd[k] = torch.full((64,), float(meta["qk_gain"]))
    dit = DiffusionTransformer3D(**cfg)
    dit.load_state_dict(sd, assign=True)
    dit.engine(dev)
    del sd
    parallelize_dit(dit, rank, world, device=dev, cfg_parallel=args.cfg_parallel)
    if args.slices > 1:
        dit.set_option("sp_slices", args.slices)
    if args.ulysses:
        dit.set_option("sp_mode", 1)
    if args.graph:
        dit.set_graph(True)
    assert dit.get_option("ipc_ranks") == (world // 2 if args.cfg_parallel else world), dit.get_option("ipc_ranks")
    assert dit.get_option("rccl_ranks") == -1          # a communicator, and not RCCL's

    T, H, W = c["latent"]
    g = torch.Generator().manual_seed(c["xseed"])
    te = {"text_embeds": torch.randn(c["L"], 3584, generator=g).to(dev), "pooled_embed": torch.randn(1, 768, generator=g).to(dev)}
    ne = {"text_embeds": torch.randn(c["Lnull"], 3584, generator=g).to(dev), "pooled_embed": torch.randn(1, 768, generator=g).to(dev)}
    pos = [torch.arange(T), torch.arange(H // 2), torch.arange(W // 2)]
    noise = torch.randn(T, H, W, 16, generator=torch.Generator().manual_seed(c["seed"]))
    sig = sigma_schedule(c["steps"], c["s"]).tolist()
    lat = noise.clone().to(dev)
    dist.barrier()
    if args.die_before_sample == rank:
        os._exit(0)          # no teardown: the peers must find out by themselves
    dit.sample(lat, sig, te, ne, pos, torch.arange(c["L"]), torch.arange(c["Lnull"]), w, scale_factor=(1.0, 2.0, 2.0), sparse_params=sparse)
    torch.cuda.synchronize(dev)
    errs = dit.get_option("ipc_errors")
    info = {"rank": rank, "world": world, "pid": os.getpid(), "device": str(dev), "ipc_ranks": dit.get_option("ipc_ranks"),
            "ipc_pair_ranks": dit.get_option("ipc_pair_ranks"), "ipc_collectives": dit.get_option("ipc_collectives"),
            "ipc_pulled_mb": dit.get_option("ipc_pulled_mb"), "ipc_flags_finegrained": dit.get_option("ipc_flags_finegrained"), "ipc_errors": errs, "cfg_branch": dit.cfg_branch() if hasattr(dit, "cfg_branch") else None}
    os.makedirs(args.out, exist_ok=True)
    torch.save(lat.cpu(), os.path.join(args.out, f"latent_rank{rank}.pt"))
    # the bench's rank_check (bench.py): every rank applies the same Euler update to the same gathered velocity -> bit-identical latents
    cs = torch.stack([lat.double().sum(), lat.double().abs().sum(), lat.view(torch.int32).sum(dtype=torch.int64).double()]).cpu()
    allcs = [torch.empty_like(cs) for _ in range(world)]
    dist.all_gather(allcs, cs)
    infos = [None] * world
    dist.all_gather_object(infos, info)
    if rank == 0:
        same = all(torch.equal(allcs[0], x) for x in allcs)
        json.dump({"rank_check": {"latent_checksums_identical_on_all_ranks": bool(same), "per_rank": [[float(v) for v in x.tolist()] for x in allcs]},
                   "ranks": infos}, open(os.path.join(args.out, "rank_check.json"), "w"), indent=1)
    dist.barrier()
    dit._destroy_engine(force=True)
    dist.destroy_process_group()
    if errs:
        raise SystemExit(f"rank {rank}: an IPC flag wait timed out (0x{errs & 0xffffffff:08x})")


if __name__ == "__main__":
    main()
