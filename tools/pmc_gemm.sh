#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> M N K [env assignments...]  — SQ counters of the GEMM kernel, per wave and K-tile
TAG=$1; M=$2; N=$3; K=$4; shift 4
export TMPDIR=/tmp; R=$PWD; cd /tmp
env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_$TAG -o p -- python $R/tools/gemm_only.py $M $N $K 2 > /dev/null 2>&1
env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_${TAG}b -o p -- python $R/tools/gemm_only.py $M $N $K 2 > /dev/null 2>&1
cd $R
python - <<PY
import csv, collections
for suf in ("", "b"):
    acc = collections.defaultdict(dict); names = {}
    try:
        rows = csv.DictReader(open("gpurun_out/pmcg_${TAG}%s/p_counter_collection.csv" % suf))
    except FileNotFoundError:
        print("no counters", suf); continue
    name = ""
    for r in rows:
        if "gemm_bf16" in r["Kernel_Name"]:
            names[r["Dispatch_Id"]] = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]
            acc[r["Dispatch_Id"]][r["Counter_Name"]] = acc[r["Dispatch_Id"]].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    big = max(acc, key=lambda i: max(acc[i].values()))   # the main launch (a tail-split GEMM also has a small 128x128 launch)
    d = acc[big]
    print("$TAG", names[big], {k: round(v, 0) for k, v in sorted(d.items())})
PY
