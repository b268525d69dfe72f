"""Summaries of tools/profile_round.sh for profiles/ (markdown + the traffic json bench.py reads)."""
import collections
import csv
import glob
import json
import os
import sys

OUT = "gpurun_out"


def stats_md(src, dst, title, top=22):
    files = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no kernel stats in", src)
        return None
    rows = list(csv.DictReader(open(files[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats --output-format csv`; total GPU kernel time {tot / 1e6:.1f} ms\n\n"
                "| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for r in rows[:top]:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("|", "/")
            name = name if len(name) <= 96 else name[:93] + "..."
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    return rows


def counters(d, want=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if want and r["Counter_Name"] not in want:
                continue
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
    return acc, names


# ---- self-attention of one block from the kernel trace: the launches bench.py's roofline sums ----
ATTN_TPARAMS = ("BOUNDED", "SPARSE", "RANGE", "PRE", "QN", "HALF")   # attn_fwd_kernel<...> template parameters, in order (attn_fwd.hip)


def attn_template_args(name):
    """'void (anonymous namespace)::attn_fwd_kernel<true, false, true, true, false, false>(AttnP)' -> {'BOUNDED': True, ...} by
    POSITION (missing trailing parameters take their defaults: false); None for anything that is not attn_fwd_kernel."""
    import re
    m = re.search(r"attn_fwd_kernel<([^>]*)>", name)
    if not m:
        return None
    raw = [v.strip() for v in m.group(1).split(",")]
    vals = [v == "true" for v in raw[:5]]
    vals += [False] * (5 - len(vals))
    gr = int(raw[5]) if len(raw) > 5 and raw[5].lstrip("-").isdigit() else (2 if len(raw) > 5 and raw[5] == "true" else 4)   # GR (round 4); 'true' = round 2-3's HALF
    d = dict(zip(ATTN_TPARAMS[:5], vals))
    d["GR"] = gr
    d["HALF"] = gr != 4
    d["QT"] = int(raw[6]) if len(raw) > 6 and raw[6].isdigit() else 2    # 16-query tiles per wave (round 4: 4 = 64-row waves, opt-in)
    return d


def attention_block_summary(trace_rows, tokens=47616, heads=28):
    """trace_rows: dicts of a rocprofv3 kernel-trace CSV.  Groups the attention launches by (kernel, grid), labels them from the
    template arguments — PRE (pre-scaled keys) = the visual self-attention, everything else cross / text attention —, and derives the
    time of one block's self-attention launch group (fixed + online forms, tail parts, merge) = all self-attention time / merges."""
    groups = collections.defaultdict(list)
    for r in trace_rows:
        n = r["Kernel_Name"]
        if "attn_fwd" in n or "attn_merge" in n:
            short = n.replace("(anonymous namespace)::", "")
            key = (short, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
            groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    table, self_total_us, n_merge = [], 0.0, 0
    for (name, wgs), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        ta = attn_template_args(name)
        if "attn_merge" in name:
            role = "merge of the split tail jobs (1 per block)"; self_total_us += sum(v); n_merge += len(v)
        elif ta and ta["PRE"]:
            form = "fixed-offset form" if ta["BOUNDED"] else "online-max form (same grid; exits at once unless the head is flagged)"
            kind = "NABLA list-driven" if ta["SPARSE"] else "dense"
            role = f"visual self-attention, {kind}, {form}" + (", fused query norm" if ta["QN"] else "") + (", 128-query workgroups" if ta["HALF"] else "")
            self_total_us += sum(v)
        else:
            role = "cross / text attention (unscaled keys)"
        table.append((name, wgs, len(v), avg, role))
    block_us = self_total_us / n_merge if n_merge else None
    tflops = 4.0 * tokens * tokens * 64 * heads / (block_us * 1e-6) / 1e12 if block_us else None
    return {"table": table, "self_total_us": self_total_us, "blocks": n_merge, "block_us": block_us, "tflops": tflops}


def attention_block_md(tag):
    files = glob.glob(f"{OUT}/prof_{tag}/**/*kernel_trace.csv", recursive=True)
    if not files:
        return
    sm = attention_block_summary(csv.DictReader(open(files[0])))
    lines = ["", "## Attention launches by grid (kernel trace of the same run)", "",
             "| kernel | workgroups | launches | avg us | role |", "|---|---:|---:|---:|---|"]
    for name, wgs, n, avg, role in sm["table"]:
        lines.append(f"| `{name[:72]}` | {wgs} | {n} | {avg:.1f} | {role} |")
    lines.append("")
    if sm["blocks"]:
        lines.append(f"Self-attention of one block = {sm['block_us'] / 1e3:.3f} ms (all visual self-attention rows above + merges: "
                     f"{sm['self_total_us'] / 1e3:.1f} ms over {sm['blocks']} blocks) -> DERIVED_TFLOPS = {sm['tflops']:.1f} TFLOP/s; bench.py's "
                     "`roofline.avg_launch_ms` is the HIP-event time of the same launch group (events also see the gaps between its launches).")
    open(f"{OUT}/{tag}_bench_kernel_stats.md", "a").write("\n".join(lines) + "\n")



def main(tag):
    rows = stats_md(f"{OUT}/prof_{tag}", f"{OUT}/{tag}_bench_kernel_stats.md",
                    f"bench.py --steps 2 --warmup 1 (config_5s_nocfg, 1x MI355X): rocprofv3 kernel stats")
    stats_md(f"{OUT}/prof_{tag}_vae", f"{OUT}/{tag}_vae_kernel_stats.md", "HunyuanVideo VAE decode of one 5 s clip (tools/vae_bench.py): rocprofv3 kernel stats")
    stats_md(f"{OUT}/prof_{tag}_nabla", f"{OUT}/{tag}_nabla_kernel_stats.md",
             "bench.py --workload 10s_nabla --steps 1 --warmup 1 (config_10s_sft latent, 93 696 tokens, NABLA P = 0.9, 1x MI355X): rocprofv3 kernel stats")

    attention_block_md(tag)

    # ---- HBM traffic per launch (FETCH_SIZE x2 on gfx950, KB units) ----
    traffic = {}
    per_kernel = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc, names = counters(f"{OUT}/pmc_{tag}_{c}", {c})
        for d, v in acc.items():
            per_kernel[names[d]][c].append(v[c])
    md = ["# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 per MI355X_MICROARCH.md; KB -> MB)", "",
          "| kernel | launches | fetch MB (x2) | write MB |", "|---|---:|---:|---:|"]
    attn_main = None
    for k, d in sorted(per_kernel.items(), key=lambda kv: -max(kv[1]["FETCH_SIZE"] or [0])):
        if not any(t in k for t in ("attn_fwd", "attn_merge", "gemm_bf16", "ln_kernel", "rmsnorm")):
            continue
        f = max(d["FETCH_SIZE"]) if d["FETCH_SIZE"] else 0.0
        w = max(d["WRITE_SIZE"]) if d["WRITE_SIZE"] else 0.0
        short = k.replace("(anonymous namespace)::", "")[:80]
        md.append(f"| `{short}` | {len(d['FETCH_SIZE'])} | {2 * f / 1024:.1f} | {w / 1024:.1f} |")
        ta = attn_template_args(k)
        if ta and ta["BOUNDED"] and ta["PRE"] and not ta["SPARSE"] and (attn_main is None or f > attn_main[1]):   # the dense fixed-offset main launch
            attn_main = (k, f, w)
    open(f"{OUT}/{tag}_hbm_traffic.md", "w").write("\n".join(md) + "\n")
    if attn_main:
        k, f, w = attn_main
        N, H = 47616, 28
        json.dump({"kernel": k.replace("(anonymous namespace)::", "")[:60], "tokens": N, "heads": H, "fetch_size_kb_raw": f,
                   "fetch_bytes_x2_gfx950": 2 * f * 1024, "write_bytes": w * 1024, "bytes_per_launch": 2 * f * 1024 + w * 1024,
                   "algorithmic_bytes": 4 * N * H * 64 * 2,
                   "source": "tools/profile_round.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes), FETCH x2 per "
                             "MI355X_MICROARCH.md; fabric-side counter: Infinity-Cache hits included"},
                  open(f"{OUT}/{tag}_attention_traffic.json", "w"))

    # ---- attention issue / wait counters ----
    md = ["# Attention kernel SQ counters (BOUNDED=1 PRESCALED=1 tools/attn_only.py: N = 47 616 tokens, 28 heads, RMS-normalised heads; last dispatch of the main launch)", ""]
    for sub in ("sq", "lds"):
        acc, names = counters(f"{OUT}/pmc_{tag}_{sub}")
        sel = [d for d in acc if "attn_fwd" in names[d]]
        if sel:
            big = max(sel, key=lambda i: max(acc[i].values()))   # the main launch (a balanced call also has small tail launches)
            sel = [big]
            d = acc[sel[-1]]
            md.append("`" + names[sel[-1]].replace("(anonymous namespace)::", "")[:70] + "`")
            md += [f"* {k}: {v:.0f}" for k, v in sorted(d.items())]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
                md.append(f"* MFMA instructions {d.get('SQ_INSTS_MFMA', 0):.0f}, VALU instructions {d.get('SQ_INSTS_VALU', 0):.0f} "
                          f"(ratio {d.get('SQ_INSTS_VALU', 0) / max(d.get('SQ_INSTS_MFMA', 1), 1):.2f} VALU per MFMA)")
            md.append("")
    open(f"{OUT}/{tag}_attention_pmc.md", "w").write("\n".join(md) + "\n")
    for fn in sorted(glob.glob(f"{OUT}/{tag}_*")):
        print(fn, os.path.getsize(fn))


if __name__ == "__main__":
    main(sys.argv[1])
