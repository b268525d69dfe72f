"""Tile-level timeline of gemm_bf16_w4_kernel on the model's shapes (needs lib/variants/libk5_w4trace.so: tools/build_variant.sh
w4trace gemm_bf16.hip -DW4_TRACE).  Every workgroup stamps the 100-MHz wall clock at tile start | K loop done | epilogue done;
printed: per shape the launch time, the mean K-loop and epilogue durations, how far apart the workgroups' epilogues of the same round
start (lockstep = all of them hit HBM at once), and the time the slowest workgroup of a round spends beyond the mean.

    python tools/gemm_w4_trace.py            (prints one block per shape; JSON lines with --json)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIMES_ONLY = "--times" in sys.argv        # the regular library, call times only (A/B of launch policies through the environment)
if not TIMES_ONLY:
    os.environ.setdefault("K5_LIB", os.path.join(ROOT, "kandinsky-5_amd", "lib", "variants", "libk5_w4trace.so"))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch  # noqa: E402

NT = 192
trace = torch.zeros(256 * NT, dtype=torch.int64, device="cuda")
os.environ["K5_GEMM_TRACE"] = "%x" % trace.data_ptr()
from kandinsky import _engine as E  # noqa: E402

N_TOK = 47616
SHAPES = [  # name, M, N, K, epilogue
    ("q|k (bias)", N_TOK, 3584, 1792, "bias"),
    ("V^T (bias per row)", 1792, N_TOK, 1792, "bias_m"),
    ("out + gate", N_TOK, 1792, 1792, "gate"),
    ("FF1 + GELU", N_TOK, 7168, 1792, "gelu"),
    ("FF2 + gate", N_TOK, 1792, 7168, "gate"),
]
if "--plain" in sys.argv:      # the model's shapes with the plain bf16 store (what a vendor-library call does)
    SHAPES = [("q|k shape, plain store", N_TOK, 3584, 1792, "none"), ("out shape, plain store", N_TOK, 1792, 1792, "none"),
              ("FF1 shape, plain store", N_TOK, 7168, 1792, "none"), ("FF2 shape, plain store", N_TOK, 1792, 7168, "none")]
if "--ablate" in sys.argv:     # which part of an epilogue's time is the shape (output row stride, bytes) and which the arithmetic
    SHAPES = [("FF1 shape, plain store", N_TOK, 7168, 1792, "none"), ("FF1 shape, GELU", N_TOK, 7168, 1792, "gelu"),
              ("q|k shape, GELU", N_TOK, 3584, 1792, "gelu"), ("q|k shape, bias", N_TOK, 3584, 1792, "bias"),
              ("out shape, bias only", N_TOK, 1792, 1792, "bias"), ("out shape, gate", N_TOK, 1792, 1792, "gate")]


def run(name, M, N, K, epi):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    if "--zero" in sys.argv:    # zero operands: the same instruction stream at the clock an idle-power chip gives it
        a.zero_(); w.zero_()
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    bias = torch.randn(M if epi == "bias_m" else N, device="cuda")
    gate = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    kw = {"bias": dict(bias=bias, epilogue=E.EPI_BIAS), "bias_m": dict(bias=bias, epilogue=E.EPI_BIAS_M),
          "gelu": dict(bias=None, epilogue=E.EPI_GELU), "none": dict(bias=None, epilogue=E.EPI_BIAS), "gate": dict(bias=bias, epilogue=E.EPI_GATE, resid=out, gate=gate)}[epi]
    if epi == "gate":
        out.copy_(resid)
    for _ in range(3):
        E.gemm(a, w, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        E.gemm(a, w, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    if TIMES_ONLY:
        return {"shape": name, "M": M, "N": N, "K": K, "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9}
    raw = trace.cpu().reshape(256, NT)
    ticks = (raw[:, NT - 1] - raw[:, NT - 2]).double()            # s_memtime: shader-clock ticks between the first tile's start and the last tile's end
    t = raw.reshape(256, NT // 3, 3).double() * 10.0      # ns
    t[:, NT // 3 - 1] = 0
    ntile = int((t[:, :, 0] > 0).sum(1).max())
    t = t[:, :ntile]
    valid = t[:, :, 0] > 0
    t0 = t[:, 0, 0][valid[:, 0]].min()
    loop, epi_t = (t[:, :, 1] - t[:, :, 0])[valid], (t[:, :, 2] - t[:, :, 1])[valid]
    nk = K // 64
    res = {"shape": name, "M": M, "N": N, "K": K, "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9, "tiles_per_wg": ntile,
           "k_loop_us": float(loop.mean()) / 1e3, "k_tile_ns": float(loop.mean()) / nk, "epilogue_us": float(epi_t.mean()) / 1e3,
           "epilogue_us_max": float(epi_t.max()) / 1e3, "epilogue_frac": float(epi_t.sum() / (loop.sum() + epi_t.sum())),
           "kernel_span_us": float((t[:, :, 2][valid].max() - t0) / 1e3)}
    span = torch.stack([t[w, int(valid[w].sum()) - 1, 2] - t[w, 0, 0] for w in range(256)])        # ns, per workgroup
    res["shader_clock_ghz"] = float((ticks / span).mean())
    res["shader_clock_ghz_minmax"] = [float((ticks / span).min()), float((ticks / span).max())]
    res["k_tile_cycles"] = res["k_tile_ns"] * res["shader_clock_ghz"]
    res["per_xcd_k_tile_ns"] = [float((t[x::8, :, 1] - t[x::8, :, 0])[valid[x::8]].mean()) / nk for x in range(8)]
    rounds = []
    for r in range(ntile):
        v = valid[:, r]
        if v.sum() < 8:
            continue
        s1, s2 = t[:, r, 1][v], t[:, r, 2][v]
        rounds.append({"round": r, "wgs": int(v.sum()), "k_tile_ns": float((s1 - t[:, r, 0][v]).mean()) / nk, "epilogue_us": float((s2 - s1).mean()) / 1e3,
                       "gap_to_next_us": float((t[:, r + 1, 0][valid[:, r + 1]] - t[:, r, 2][valid[:, r + 1]]).mean()) / 1e3 if r + 1 < ntile else 0.0,
                       "epilogue_start_spread_us": float(s1.max() - s1.min()) / 1e3,
                       "epilogue_start_std_us": float(s1.std()) / 1e3, "end_spread_us": float(s2.max() - s2.min()) / 1e3})
    res["rounds"] = rounds
    return res


def main():
    as_json = "--json" in sys.argv
    for sh in SHAPES:
        r = run(*sh)
        if as_json or TIMES_ONLY:
            print(json.dumps(r))
            continue
        print(f"{r['shape']:22s} {r['M']}x{r['N']}x{r['K']}: {r['ms'] * 1e3:7.1f} us = {r['tflops']:6.0f} TFLOP/s; {r['tiles_per_wg']} tiles per workgroup; "
              f"K loop {r['k_loop_us']:.1f} us ({r['k_tile_ns']:.0f} ns per K-tile), epilogue {r['epilogue_us']:.2f} us (max {r['epilogue_us_max']:.2f}) = "
              f"{100 * r['epilogue_frac']:.1f} % of a tile; main launch span {r['kernel_span_us']:.1f} us; shader clock {r['shader_clock_ghz']:.3f} GHz ({r['shader_clock_ghz_minmax'][0]:.3f}-{r['shader_clock_ghz_minmax'][1]:.3f}) "
              f"=> {r['k_tile_cycles']:.0f} cycles per K-tile (128 MFMA x 16 = 2048)")
        print("      per-XCD ns per K-tile:", " ".join(f"{v:.0f}" for v in r["per_xcd_k_tile_ns"]))
        for rd in r["rounds"]:
            print(f"      round {rd['round']}: {rd['wgs']} workgroups, {rd['k_tile_ns']:.0f} ns per K-tile, epilogue {rd['epilogue_us']:.2f} us, epilogues start within {rd['epilogue_start_spread_us']:.2f} us (std {rd['epilogue_start_std_us']:.2f}), "
                  f"tiles end within {rd['end_spread_us']:.2f} us")


if __name__ == "__main__":
    main()
