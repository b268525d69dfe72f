#!/bin/bash
# LDS bank-conflict share per kernel over the DiT step (2 blocks), the NABLA step and the VAE decode: tools/pmc_lds_conflicts.sh TAG
TAG=${1:-r05}; export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmcl_${TAG}_dit -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-breakdown --no-parity-check --blocks 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmcl_${TAG}_nabla -o p -- python $R/bench.py --workload 10s_nabla --steps 1 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-breakdown --blocks 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmcl_${TAG}_vae -o p -- python $R/tools/vae_bench.py > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections
print("# LDS bank conflicts per kernel (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS; sums over all launches)\n")
for part in ("dit", "nabla", "vae"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(f"gpurun_out/pmcl_${TAG}_{part}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]][r["Counter_Name"]] += float(r["Counter_Value"])
    print(f"## {part}\n\n| kernel | LDS instructions | LDS-array cycles | conflict cycles | share |\n|---|---:|---:|---:|---:|")
    for k, d in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"]):
        if d["SQ_LDS_IDX_ACTIVE"] < 1e5: continue
        print(f"| \`{k}\` | {d['SQ_INSTS_LDS']:.0f} | {d['SQ_LDS_IDX_ACTIVE']:.0f} | {d['SQ_LDS_BANK_CONFLICT']:.0f} | {d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.2f} |")
    print()
PY
