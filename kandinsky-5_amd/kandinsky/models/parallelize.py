"""Multi-GPU plan — replaces the reference's DTensor tensor-parallel plan (kandinsky/models/parallelize.py:11-102).

The reference shards attention HEADS and all-reduces the full (N, 1792) activation three times per block
(SURVEY.md §5.8).  Here the TOKEN axis is sharded instead: rank r owns a contiguous, 64-aligned slice of the
N visual tokens; every per-token op (AdaLN, projections, RMSNorm, RoPE, cross-attention, feed-forward) runs on
the local slice only, and the one exchange per block is an in-place all-gather of K and V^T (RCCL over xGMI),
after which each rank attends its own query rows to all keys.  No head-count constraint (28 heads do not divide
by 8).  The arithmetic lives in the engine (csrc/engine.hip, run_self_attention_sp); this module holds the
host-side shard bookkeeping shared with the tests.
"""


def token_shard(num_tokens: int, world: int, rank: int):
    """(start, count) of the token rows owned by `rank`.  The engine requires equal, 64-aligned shards."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    if num_tokens % (64 * world):
        raise ValueError(f"sequence parallel x{world} needs the token count ({num_tokens}) to be a multiple of {64 * world}")
    n = num_tokens // world
    return rank * n, n


def parallelize_dit(model, rank: int, world: int, device=None):
    """Drop-in for the reference's parallelize_dit(model, tp_mesh): enables sequence parallelism on the engine."""
    if world > 1:
        model.enable_sequence_parallel(rank, world, device=device)
    return model
