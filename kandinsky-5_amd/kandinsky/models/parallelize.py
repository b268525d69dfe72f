"""Multi-GPU plan — replaces the reference's DTensor tensor-parallel plan (kandinsky/models/parallelize.py:11-102).

The reference shards attention HEADS and all-reduces the full (N, 1792) activation three times per block
(SURVEY.md §5.8).  Here the TOKEN axis is sharded instead: rank r owns a contiguous, 64-aligned slice of the
N visual tokens; every per-token op (AdaLN, projections, RMSNorm, RoPE, cross-attention, feed-forward) runs on
the local slice only, and the one exchange per block is an in-place all-gather of K and V^T (RCCL over xGMI),
after which each rank attends its own query rows to all keys.  No head-count constraint (28 heads do not divide
by 8).  The arithmetic lives in the engine (csrc/engine.hip, run_self_attention_sp); this module holds the
host-side shard bookkeeping shared with the tests.
"""


def token_shard(num_tokens: int, world: int, rank: int, slices: int = 1):
    """(start, count) of the token rows owned by `rank` — the engine's layout (csrc/engine.hip forward_impl): whole 64-token
    blocks, ceil(blocks / world) per rank, the LAST rank takes what is left (3660 blocks over 8 ranks = 7 x 458 + 454), so the
    slot size in the gather buffers is the same on every rank and only the tail of the last slot is unused.  With the sliced
    K / V^T exchange (engine option "sp_slices" = slices > 1) a slot is a whole number of slices: a multiple of 64 * slices."""
    if world < 1 or not 0 <= rank < world or slices < 1:
        raise ValueError("bad rank/world")
    if num_tokens % 64:
        raise ValueError(f"sequence parallelism needs whole 64-token blocks (token count {num_tokens})")
    slot = shard_slot(num_tokens, world, slices)
    if (world - 1) * slot >= num_tokens:
        raise ValueError(f"sequence parallel x{world}: {num_tokens // 64} token blocks leave a rank without work")
    start = rank * slot
    return start, min(slot, num_tokens - start)


def shard_slot(num_tokens: int, world: int, slices: int = 1) -> int:
    """rows of one rank's slot in the gather buffers (>= every rank's own count)"""
    return -(-(num_tokens // 64) // (world * slices)) * slices * 64


class ParallelLayout:
    """Rank layout of one node.  Without CFG-parallel: one sequence-parallel group of `world` ranks.  With it
    (SURVEY.md §8e "CFG-parallel"): two groups of world/2 ranks — ranks [0, world/2) run the conditional forward, ranks
    [world/2, world) the unconditional one, each group sequence-parallel inside — and rank i of one group is paired with
    rank i of the other for the single velocity exchange per step."""

    def __init__(self, rank: int, world: int, cfg_parallel: bool = False):
        if world < 1 or not 0 <= rank < world:
            raise ValueError("bad rank/world")
        if cfg_parallel and world % 2:
            raise ValueError(f"CFG-parallel needs an even number of ranks, got {world}")
        self.rank, self.world, self.cfg_parallel = rank, world, bool(cfg_parallel)
        self.sp_world = world // 2 if cfg_parallel else world
        self.branch = rank // self.sp_world if cfg_parallel else 0       # 0 = conditional, 1 = unconditional
        self.sp_rank = rank % self.sp_world
        self.sp_groups = [list(range(b * self.sp_world, (b + 1) * self.sp_world)) for b in range(2 if cfg_parallel else 1)]
        self.pair_groups = [[i, i + self.sp_world] for i in range(self.sp_world)] if cfg_parallel else []

    @property
    def sp_ranks(self):
        return self.sp_groups[self.branch]

    @property
    def pair_ranks(self):
        return self.pair_groups[self.sp_rank] if self.cfg_parallel else [self.rank]


def make_groups(layout: ParallelLayout):
    """torch.distributed sub-groups of the layout.  Collective: every rank creates every group, in the same order."""
    import torch.distributed as dist
    sp = pair = None
    for ranks in layout.sp_groups:
        g = dist.new_group(ranks) if len(ranks) < layout.world or layout.cfg_parallel else None
        if layout.rank in ranks:
            sp = g
    for ranks in layout.pair_groups:
        g = dist.new_group(ranks)
        if layout.rank in ranks:
            pair = g
    return sp, pair


def exchange_velocity(v_mine, pair_group, out=None):
    """CFG-parallel: one all-gather over the 2-rank pair -> (v_cond, v_uncond) (pair rank order = branch order)."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty((2,) + tuple(v_mine.shape), dtype=v_mine.dtype, device=v_mine.device)
    # output passed as the dim-0 concatenation (the form every backend accepts)
    dist.all_gather_into_tensor(out.view((2 * v_mine.shape[0],) + tuple(v_mine.shape[1:])), v_mine.contiguous(), group=pair_group)
    return out[0], out[1]


def parallelize_dit(model, rank: int, world: int, device=None, cfg_parallel: bool = False):
    """Drop-in for the reference's parallelize_dit(model, tp_mesh): sequence parallelism inside the engine (RCCL), and
    optionally the conditional / unconditional forwards of classifier-free guidance on two disjoint rank groups."""
    if world <= 1:
        return model
    layout = ParallelLayout(rank, world, cfg_parallel)
    sp_group, pair_group = make_groups(layout) if cfg_parallel else (None, None)
    if layout.sp_world > 1:
        model.enable_sequence_parallel(layout.sp_rank, layout.sp_world, device=device, group=sp_group, src=layout.sp_ranks[0])
    model._layout = layout
    model._cfg_parallel = (layout.branch, pair_group) if cfg_parallel else None
    if cfg_parallel and hasattr(model, "enable_cfg_pair"):
        # the exchange lives in the engine: `generate` stays on the fused k5_sample loop (and its captured step) — the pair's torch
        # group only carries the 128-byte RCCL id
        model.enable_cfg_pair(layout.branch, group=pair_group, src=layout.pair_ranks[0], device=device)
    if getattr(model, "mag_ratios", None) is not None:   # MagCache slot bookkeeping depends on the branch
        from ..magcache_utils import _apply
        _apply(model)
    return model
