#!/bin/bash
# usage: tools/pmc_attn.sh <tag> [env assignments...]   — SQ wave-state counters for the attention kernel
TAG=$1; shift
export TMPDIR=/tmp; R=$PWD; cd /tmp
env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o p -- python $R/tools/attn_only.py 2 > /dev/null 2>&1
env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}b -o p -- python $R/tools/attn_only.py 2 > /dev/null 2>&1
cd $R
python - <<PY
import csv, collections
for suf in ("", "b"):
    acc = collections.defaultdict(dict)
    try:
        rows = csv.DictReader(open("gpurun_out/pmc_${TAG}%s/p_counter_collection.csv" % suf))
    except FileNotFoundError:
        continue
    for r in rows:
        if "attn_fwd" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] = acc[r["Dispatch_Id"]].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    d = list(acc.values())[-1]
    nwt = 5208 * 8 * 744.0
    print("$TAG", {k: round(v / nwt, 1) for k, v in sorted(d.items())}, "(per wave-tile)")
PY
