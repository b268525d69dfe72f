#!/bin/bash
# SQ counters of the list-driven (NABLA) attention kernel at its operating density, next to the dense kernel's ("none is
# committed").  usage: tools/pmc_sparse.sh <tag> [nabla_p]      -> gpurun_out/<tag>_sparse_attention_pmc.md
TAG=${1:-r04}; NP=${2:-0.0}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp
CMD="python $R/bench.py --workload 10s_nabla --nabla-p $NP --steps 1 --warmup 1 --blocks 2 --no-cpu-baseline --no-vae --no-breakdown --no-parity-check"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmcs_${TAG}a -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmcs_${TAG}b -o p -- $CMD > /dev/null 2>&1
cd $R
python - <<PY
import csv, collections, glob
tot = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for suf in "ab":
    for fn in glob.glob("gpurun_out/pmcs_${TAG}%s/**/*counter_collection.csv" % suf, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "attn_fwd_kernel" in r["Kernel_Name"]:
                tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Kernel_Name"]].add((suf, r["Dispatch_Id"]))
md = ["# SQ counters of the attention instantiations in \`bench.py --workload 10s_nabla --nabla-p $NP --blocks 2\` (93 696 tokens; rocprofv3 --pmc, two passes; sums over every dispatch of the run)", ""]
for name, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0)):
    m = c.get("SQ_INSTS_MFMA", 0)
    if m < 1e6: continue
    md.append(f"\`{name}\` ({len(nd[name]) // 2} dispatches)")
    for k in sorted(c): md.append(f"* {k}: {c[k]:.0f}" + (f"  ({c[k] / m:.3f} per MFMA)" if k != "SQ_INSTS_MFMA" else ""))
    w = c.get("SQ_WAVE_CYCLES", 0)
    if w: md.append(f"* derived: MFMA-busy cycles per wave cycle x 4 (quad-cycle units) = {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w / 4:.3f}; waiting fraction of wave cycles {c.get('SQ_WAIT_INST_ANY', 0) / w:.3f}; LDS-wait {c.get('SQ_WAIT_INST_LDS', 0) / w:.3f}")
    md.append("")
open("gpurun_out/${TAG}_sparse_attention_pmc.md", "w").write("\n".join(md) + "\n")
print("\n".join(md))
PY
