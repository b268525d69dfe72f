#!/bin/bash
# usage: tools/pmc_attn2.sh <tag> [env assignments...] — VALU / MFMA / LDS counters of the (bounded) attention kernel
TAG=$1; shift
export TMPDIR=/tmp; R=$PWD; cd /tmp
env BOUNDED=1 PRESCALED=1 "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmca_$TAG -o p -- python $R/tools/attn_only.py 2 > /dev/null 2>&1
env BOUNDED=1 PRESCALED=1 "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmca_${TAG}b -o p -- python $R/tools/attn_only.py 2 > /dev/null 2>&1
cd $R
python - <<PY
import csv, collections, glob
for suf in ("", "b"):
    acc = collections.defaultdict(dict); names = {}
    for fn in glob.glob("gpurun_out/pmca_${TAG}%s/**/*counter_collection.csv" % suf, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "attn_fwd" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] = acc[r["Dispatch_Id"]].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
    if not acc: print("no data"); continue
    k = max(acc, key=lambda d: acc[d].get("SQ_INSTS_MFMA", acc[d].get("SQ_WAVE_CYCLES", 0)))
    nwt = 5208 * 8 * 744.0
    print("$TAG", names[k][:48], {kk: round(v / nwt, 1) for kk, v in sorted(acc[k].items())}, "(per wave-tile)")
PY
