"""Multi-GPU path.  CPU: the token-sharded algorithm (local projections, K / V^T all-gather, local-query
attention, velocity all-gather) run by 2 gloo ranks on the oracle arithmetic equals the unsharded forward.
GPU (one MI355X): a world=1 RCCL communicator drives the engine's sharded code path end to end and must
reproduce the fused single-GPU path bit for bit."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import k5_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _all_to_all(send, rank, world):
    """block g of `send` goes to rank g; returns the blocks received (gloo has no alltoall: every rank gathers everybody's send list and
    keeps the blocks addressed to it — the same data movement as far as the receiver can tell)"""
    mine = torch.stack(send, 0)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    return [everyone[p][rank] for p in range(world)]


def _sp_forward(sd, cfg, x, text, pooled, time, vpos, tpos, rank, world, ulysses=False):
    """Sequence-parallel restatement of O.dit_forward: mirrors csrc/engine.hip forward_impl + run_self_attention_sp."""
    from kandinsky.models.parallelize import shard_slot, token_shard
    mode = "fp32"
    txt = O.text_embeddings(sd, "text_embeddings", text, mode)
    temb = O.time_embeddings(sd, time, cfg) + O.text_embeddings(sd, "pooled_text_embeddings", pooled, mode)
    vis = O.visual_embeddings(sd, x, cfg, mode)
    ta = O.rope_1d_args(tpos, cfg.head_dim)
    for i in range(cfg.num_text_blocks):
        txt = O.encoder_block(sd, f"text_transformer_blocks.{i}", txt, temb, torch.cos(ta), torch.sin(ta), cfg, mode)
    Tp, Hp, Wp, D = vis.shape
    va = O.rope_3d_args((Tp, Hp, Wp), vpos, cfg.axes_dims, (1.0, 2.0, 2.0)).reshape(-1, 32)
    N = Tp * Hp * Wp
    t0, n = token_shard(N, world, rank)
    slot = shard_slot(N, world)
    vis = vis.reshape(N, D)[t0:t0 + n]
    cos, sin = torch.cos(va)[t0:t0 + n], torch.sin(va)[t0:t0 + n]
    H = cfg.num_heads
    for i in range(cfg.num_visual_blocks):
        p = f"visual_transformer_blocks.{i}"
        mod = O.modulation(sd, f"{p}.visual_modulation", temb)
        sa, ca, ff = torch.chunk(mod, 3, dim=-1)
        shift, scale, gate = torch.chunk(sa, 3, dim=-1)
        h = O.scale_shift_norm(vis, scale, shift, mode)
        q, k, v = O._attn_qkv(sd, f"{p}.self_attention", h, h, mode, H)
        q, k = O.apply_rotary(q, cos, sin, mode), O.apply_rotary(k, cos, sin, mode)
        # every rank's slot holds `slot` rows; only the last rank's may be partly unused (its tail is never read)
        if ulysses:
            # run_self_attention_ulysses: all-to-all trades this rank's rows of ALL heads for ALL rows of its H / world heads (blocks of
            # `slot` rows per rank, the last one short), one full-sequence attention per owned head, all-to-all of the outputs back
            Hg = H // world                                    # heads per rank (Hp is the patched height here)
            def to_heads(t):                                   # (n, H, 64) -> (N, Hg, 64)
                pad = torch.zeros(slot, H, 64)
                pad[:n] = t
                send = [pad[:, g * Hg:(g + 1) * Hg].contiguous() for g in range(world)]
                return torch.cat(_all_to_all(send, rank, world), 0)[:N]
            qa, ka, va_ = to_heads(q), to_heads(k), to_heads(v.reshape(n, H, 64))
            oa = O.sdpa(qa, ka, va_, mode).reshape(N, Hg, 64)
            opad = torch.zeros(world * slot, Hg, 64)
            opad[:N] = oa
            send = [opad[s_ * slot:(s_ + 1) * slot].contiguous() for s_ in range(world)]
            o = torch.cat(_all_to_all(send, rank, world), 1)[:n].reshape(n, D)   # head groups side by side = the original head order
            o = O._linear(o, sd[f"{p}.self_attention.out_layer.weight"], sd[f"{p}.self_attention.out_layer.bias"], mode)
            vis = O.gate_sum(vis, o, gate, mode)
            shift, scale, gate = torch.chunk(ca, 3, dim=-1)
            vis = O.gate_sum(vis, O.cross_attention(sd, f"{p}.cross_attention", O.scale_shift_norm(vis, scale, shift, mode),
                                                    txt, cfg, mode), gate, mode)
            shift, scale, gate = torch.chunk(ff, 3, dim=-1)
            vis = O.gate_sum(vis, O.feed_forward(sd, f"{p}.feed_forward", O.scale_shift_norm(vis, scale, shift, mode), mode),
                             gate, mode)
            continue
        kpad, vtpad = torch.zeros(slot, H, 64), torch.zeros(D, slot)
        kpad[:n], vtpad[:, :n] = k, v.reshape(n, D).t()
        kfull = [torch.empty_like(kpad) for _ in range(world)]
        vtfull = [torch.empty_like(vtpad) for _ in range(world)]
        dist.all_gather(kfull, kpad)
        dist.all_gather(vtfull, vtpad)
        kall = torch.cat(kfull, 0)[:N]
        vall = torch.cat(vtfull, 1)[:, :N].t().reshape(N, H, 64)
        o = O.sdpa(q, kall, vall, mode)
        o = O._linear(o, sd[f"{p}.self_attention.out_layer.weight"], sd[f"{p}.self_attention.out_layer.bias"], mode)
        vis = O.gate_sum(vis, o, gate, mode)
        shift, scale, gate = torch.chunk(ca, 3, dim=-1)
        vis = O.gate_sum(vis, O.cross_attention(sd, f"{p}.cross_attention", O.scale_shift_norm(vis, scale, shift, mode),
                                                txt, cfg, mode), gate, mode)
        shift, scale, gate = torch.chunk(ff, 3, dim=-1)
        vis = O.gate_sum(vis, O.feed_forward(sd, f"{p}.feed_forward", O.scale_shift_norm(vis, scale, shift, mode), mode),
                         gate, mode)
    y = O.out_layer(sd, vis, temb, cfg, mode)
    ypad = torch.zeros(slot, y.shape[1])
    ypad[:n] = y
    yall = [torch.empty_like(ypad) for _ in range(world)]
    dist.all_gather(yall, ypad)
    return O.unpatchify(torch.cat(yall, 0)[:N].reshape(Tp, Hp, Wp, -1), cfg.patch_size)


def _worker(rank, world, port, q, T=2, ulysses=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = O.DitConfig(in_visual_dim=16, in_text_dim=96, in_text_dim2=48, time_dim=64, out_visual_dim=16,
                      patch_size=(1, 2, 2), model_dim=128, ff_dim=256, num_text_blocks=1, num_visual_blocks=2,
                      axes_dims=(16, 24, 24), visual_cond=True)
    sd = O.synthetic_state_dict(cfg, seed=5, std=0.05)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(T, 16, 16, 33, generator=g)  # T*8*8 tokens: T = 2 -> 2 ranks x 64; T = 3 -> 128 + 64 (uneven shards)
    text, pooled = torch.randn(9, 96, generator=g), torch.randn(1, 48, generator=g)
    t = torch.tensor([432.0])
    vpos = [torch.arange(T), torch.arange(8), torch.arange(8)]
    out = _sp_forward(sd, cfg, x, text, pooled, t, vpos, torch.arange(9), rank, world, ulysses)
    ref = O.dit_forward(sd, cfg, x, text, pooled, t, vpos, torch.arange(9), (1.0, 2.0, 2.0), None, "fp32")
    q.put((rank, float((out - ref).abs().max()), float(ref.abs().max())))
    dist.destroy_process_group()


@pytest.mark.parametrize("T,ulysses", [(2, False), (3, False), (2, True), (3, True)])
def test_sequence_parallel_algorithm_two_ranks_gloo(T, ulysses):
    """ulysses: the all-to-all form (engine option "sp_mode" = 1; the tiny model's 2 heads = one per rank), even and uneven shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + T + (10 if ulysses else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, T, ulysses)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    for rank, err, scale in res:
        assert err <= 1e-5 * max(scale, 1.0), (rank, err, scale)


def test_token_shard_contract():
    from kandinsky.models.parallelize import token_shard
    assert token_shard(47616, 8, 3) == (3 * 5952, 5952)
    assert token_shard(47616, 1, 0) == (0, 47616)
    assert token_shard(93696, 4, 3) == (3 * 23424, 23424)
    # 3660 blocks (BASELINE config 5, 1280x768 10 s) over 8 ranks: 7 x 458 blocks + 454
    assert token_shard(234240, 8, 0) == (0, 458 * 64) and token_shard(234240, 8, 6) == (6 * 458 * 64, 458 * 64)
    assert token_shard(234240, 8, 7) == (7 * 458 * 64, 454 * 64)
    assert sum(token_shard(234240, 8, r)[1] for r in range(8)) == 234240
    # sliced K / V^T exchange ("sp_slices" = 2): slots are whole slices (multiples of 128 tokens); 744 blocks over 8 ranks = 7 x 94 + 86
    assert token_shard(47616, 8, 0, slices=2) == (0, 94 * 64) and token_shard(47616, 8, 7, slices=2) == (7 * 94 * 64, 86 * 64)
    assert sum(token_shard(47616, 8, r, slices=2)[1] for r in range(8)) == 47616
    assert token_shard(234240, 8, 7, slices=2) == (7 * 458 * 64, 454 * 64)          # 458 is already even
    assert token_shard(192, 2, 0) == (0, 128) and token_shard(192, 2, 1) == (128, 64)
    with pytest.raises(ValueError):
        token_shard(128, 2, 2)
    with pytest.raises(ValueError):
        token_shard(640, 8, 0)     # 10 blocks, ceil = 2 per rank: ranks 5..7 would idle
    with pytest.raises(ValueError):
        token_shard(100, 2, 0)     # not whole blocks


@pytest.mark.gpu
@pytest.mark.parametrize("qfuse", [0, 2])
def test_engine_sharded_path_world1_rccl_matches_fused(golden, golden_meta, tiny_sd, qfuse):
    """Bit-identical to the single-handle path when both normalise the queries in the same place ("attn_fuse_qnorm": 0 = the
    standalone pass on both, 2 = inside the attention kernel on both; the defaults differ — fused on one rank, standalone in the sharded
    schedule — and then a query element may round the other way, tests/test_gpu_loopback.py bounds that)."""
    from kandinsky.models.dit import DiffusionTransformer3D
    c = dict(golden_meta["tiny_config"])
    pos = [torch.arange(2), torch.arange(8), torch.arange(8)]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 16, 33, generator=g).cuda()
    text, pooled = torch.randn(9, 96, generator=g).cuda(), torch.randn(1, 48, generator=g).cuda()
    outs = []
    for sp in (False, True):
        dit = DiffusionTransformer3D(**c)
        dit.load_state_dict(tiny_sd, assign=True)
        dit = dit.to("cuda:0")
        dit.engine("cuda:0")
        dit.set_option("attn_fuse_qnorm", qfuse)
        if sp:
            dit.enable_sequence_parallel(0, 1, device="cuda:0")
        outs.append(dit(x, text, pooled, torch.tensor([432.0]), pos, torch.arange(9), scale_factor=(1.0, 2.0, 2.0)))
        del dit
    assert torch.equal(outs[0], outs[1])
    # and a token count that does not split is refused loudly on the sharded path
    dit = DiffusionTransformer3D(**c)
    dit.load_state_dict(tiny_sd, assign=True)
    dit = dit.to("cuda:0").enable_sequence_parallel(0, 1, device="cuda:0")
    with pytest.raises(RuntimeError, match="whole 64-token blocks"):
        dit(golden["fwd.x"].cuda(), text, pooled, torch.tensor([432.0]), [torch.arange(3), torch.arange(4), torch.arange(6)],
            torch.arange(9), scale_factor=(1.0, 2.0, 2.0))


# ------------------------------------------------------------------------------------------ CFG-parallel + VAE tile distribution
def test_parallel_layout_contract():
    from kandinsky.models.parallelize import ParallelLayout
    L = ParallelLayout(5, 8, cfg_parallel=True)     # 8 GPUs: CFG(2) x SP(4)
    assert (L.branch, L.sp_rank, L.sp_world) == (1, 1, 4)
    assert L.sp_ranks == [4, 5, 6, 7] and L.pair_ranks == [1, 5]
    assert ParallelLayout(1, 8, True).pair_ranks == [1, 5] and ParallelLayout(1, 8, True).branch == 0
    L = ParallelLayout(1, 2, cfg_parallel=True)     # 2 GPUs: one forward each, no per-block exchange at all
    assert (L.branch, L.sp_world, L.sp_ranks, L.pair_ranks) == (1, 1, [1], [0, 1])
    L = ParallelLayout(3, 4, cfg_parallel=False)
    assert (L.branch, L.sp_rank, L.sp_world, L.sp_ranks, L.pair_ranks) == (0, 3, 4, [0, 1, 2, 3], [3])
    with pytest.raises(ValueError):
        ParallelLayout(0, 3, cfg_parallel=True)


def _cfg_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from kandinsky.models.parallelize import ParallelLayout, make_groups, exchange_velocity
    cfg = O.DitConfig(in_visual_dim=16, in_text_dim=96, in_text_dim2=48, time_dim=64, out_visual_dim=16,
                      patch_size=(1, 2, 2), model_dim=128, ff_dim=256, num_text_blocks=1, num_visual_blocks=2,
                      axes_dims=(16, 24, 24), visual_cond=True)
    sd = O.synthetic_state_dict(cfg, seed=5, std=0.05)
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(2, 8, 8, 16, generator=g)
    te = {"text_embeds": torch.randn(9, 96, generator=g), "pooled_embed": torch.randn(1, 48, generator=g)}
    ne = {"text_embeds": torch.randn(4, 96, generator=g), "pooled_embed": torch.randn(1, 48, generator=g)}
    vpos = [torch.arange(2), torch.arange(4), torch.arange(4)]
    w, steps, mode = 3.0, 3, "bf16"
    layout = ParallelLayout(rank, world, cfg_parallel=True)
    _, pair = make_groups(layout)
    # the CFG-parallel loop of kandinsky/generation_utils.py:generate, on the oracle arithmetic
    mine, mine_pos = (te, torch.arange(9)) if layout.branch == 0 else (ne, torch.arange(4))
    img = noise.clone()
    sig = O.sigma_schedule(steps, 5.0)
    for i in range(steps):
        x = torch.cat([img, torch.zeros_like(img), torch.zeros(*img.shape[:-1], 1)], dim=-1)
        v = O.dit_forward(sd, cfg, x, mine["text_embeds"], mine["pooled_embed"], sig[i].unsqueeze(0) * 1000, vpos, mine_pos,
                          (1.0, 2.0, 2.0), None, mode)
        vc, vu = exchange_velocity(v.to(torch.bfloat16), pair)
        vc, vu = vc.float(), vu.float()
        v = O._r(vu + O._r(w * O._r(vc - vu, mode), mode), mode)
        img = img + O._r((sig[i + 1] - sig[i]) * v, mode)
    ref = O.generate(sd, cfg, noise, steps, te, ne, vpos, torch.arange(9), torch.arange(4), w, 5.0, (1.0, 2.0, 2.0), None, mode)
    q.put((rank, float((img - ref).abs().max()), float(ref.abs().max())))
    dist.destroy_process_group()


def _spawn(worker, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == list(range(world))
    return res


def test_cfg_parallel_two_ranks_gloo():
    """cond on rank 0, uncond on rank 1, one velocity exchange per step: both ranks end with the latent of the
    sequential two-forward sampler, bit for bit (bf16-island oracle arithmetic)."""
    for rank, err, scale in _spawn(_cfg_worker):
        assert err == 0.0, (rank, err, scale)


def _vae_worker(rank, world, port, q, nf=31):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    from oracle import vae_oracle as VO

    def fake_tile(self, z):   # deterministic stand-in for the engine's decoder (the test is about tile bookkeeping)
        t, h, w = z.shape[2:]
        up = torch.nn.functional.interpolate(z[:, :3].float(), size=(4 * (t - 1) + 1, 8 * h, 8 * w), mode="nearest")
        return (up * 0.5 + z.float().mean()).to(torch.bfloat16)

    def fake_blend(a, b, extent, dim):
        return VO.blend(a.float(), b.float().clone(), extent, dim, "bf16").to(b.dtype)

    outs = []
    for parallel in (False, True):
        with torch.device("meta"):
            vae = AutoencoderKLHunyuanVideo(block_out_channels=(64, 64, 128, 128), norm_num_groups=16)
        vae._decode_tile = fake_tile.__get__(vae)
        vae._blend = staticmethod(fake_blend).__get__(None, AutoencoderKLHunyuanVideo)
        if parallel:
            vae.enable_tile_parallel(rank, world)
        z = torch.randn(1, 16, nf, 8, 12, generator=torch.Generator().manual_seed(11))   # 31: 121 frames -> 14 temporal tiles; 7: 25 frames -> 2 tiles
        outs.append(vae.decode(z).sample)
    q.put((rank, float((outs[0].float() - outs[1].float()).abs().max()), tuple(outs[1].shape)))
    dist.destroy_process_group()


def test_vae_temporal_tiles_distributed_two_ranks_gloo():
    for rank, err, shape in _spawn(_vae_worker):
        assert err == 0.0 and shape == (1, 3, 121, 64, 96), (rank, err, shape)


def _vae_worker_short(rank, world, port, q):
    _vae_worker(rank, world, port, q, nf=7)


def test_vae_fewer_temporal_tiles_than_ranks_gloo():
    """Round 6 (found by the multi-process launch-contract test): a 1 s clip has 2 temporal tiles; on 3 (4, 8) ranks the FIRST round of the tile
    distribution is already ragged — the ranks without a tile take part in the gather with zeros of a tile's shape instead of refusing."""
    for rank, err, shape in _spawn(_vae_worker_short, world=3):
        assert err == 0.0 and shape == (1, 3, 25, 64, 96), (rank, err, shape)


def _ipc_host_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      K5_SP_TRANSPORT="ipc", K5_OVERSUBSCRIBE="1")
    from kandinsky.models.dit import _broadcast_ipc_name, sp_transport
    from kandinsky.models.parallelize import ParallelLayout, make_groups
    from kandinsky.utils import init_rank_process_group, rank_device_index
    init_rank_process_group(rank)                         # K5_SP_TRANSPORT=ipc -> a host-side (gloo) group: no device is touched
    name = _broadcast_ipc_name("sp", rank == 0, world, None, 0)
    layout = ParallelLayout(rank, world, cfg_parallel=True)
    _, pair = make_groups(layout)
    pname = _broadcast_ipc_name("pair", layout.branch == 0, 2, pair, layout.pair_ranks[0])
    q.put((rank, (dist.get_backend(), sp_transport(), name, pname, rank_device_index(rank))))
    dist.destroy_process_group()


def test_ipc_transport_host_contract_two_ranks_gloo():
    """The host half of the IPC transport (kandinsky/utils.py init_rank_process_group, models/dit.py enable_sequence_parallel): under
    K5_SP_TRANSPORT=ipc the launcher's process group is gloo (RCCL would refuse ranks that share a device), the group's rank 0 makes up
    the name of the shared-memory control block and every rank receives the same one — for the sequence-parallel group and for the CFG
    pair — and K5_OVERSUBSCRIBE wraps the ranks around the devices that exist (none here: index 0)."""
    res = dict((r, v) for r, v in _spawn(_ipc_host_worker))
    assert res[0][0] == res[1][0] == "gloo" and res[0][1] == "ipc"
    assert res[0][2] == res[1][2] and res[0][2].startswith("/k5ipc_sp_")
    assert res[0][3] == res[1][3] and res[0][3].startswith("/k5ipc_pair_") and res[0][3] != res[0][2]
    assert res[0][4] == res[1][4] == 0


def test_transport_selection_and_device_of_a_rank(monkeypatch):
    from kandinsky.models.dit import sp_transport
    from kandinsky.utils import rank_device_index
    monkeypatch.delenv("K5_SP_TRANSPORT", raising=False)
    monkeypatch.delenv("K5_OVERSUBSCRIBE", raising=False)
    assert sp_transport() == "rccl" and sp_transport("IPC") == "ipc"
    monkeypatch.setenv("K5_SP_TRANSPORT", "ipc")
    assert sp_transport() == "ipc" and sp_transport("rccl") == "rccl"
    with pytest.raises(ValueError):
        sp_transport("mpi")
    assert rank_device_index(5) == 5                       # the reference's contract: cuda:LOCAL_RANK (utils.py:40-45)
    monkeypatch.setenv("K5_OVERSUBSCRIBE", "1")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert [rank_device_index(r) for r in range(5)] == [0, 1, 0, 1, 0]
