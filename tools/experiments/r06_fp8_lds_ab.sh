#!/bin/bash
# round 6: the four-wave fp8 GEMM with conflict-free fragment reads (1088-B pieces, half swap in pieces 8..15) against round 5's layout (variant lib), timing + SQ counters
cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do
  echo "--- round 5 layout"; K5_LIB=$PWD/kandinsky-5_amd/lib/variants/libk5_fp8_r5.so python tools/gemm_fp8_time.py 2>/dev/null | grep TFLOP
  echo "--- round 6 layout"; python tools/gemm_fp8_time.py 2>/dev/null | grep TFLOP
done
for v in r5 r6; do
  L=$PWD/kandinsky-5_amd/lib/libk5.so; [ $v = r5 ] && L=$PWD/kandinsky-5_amd/lib/variants/libk5_fp8_r5.so
  (cd /tmp; K5_LIB=$L rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_fp8_$v -o p -- python /root/repo/tools/gemm_fp8_time.py > /dev/null 2>&1)
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("gpurun_out/pmc_fp8_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "gemm_fp8" in k and d["SQ_LDS_IDX_ACTIVE"] > 0: print("$v", k, "LDS cycles %.3g conflict cycles %.3g share %.2f" % (d["SQ_LDS_IDX_ACTIVE"], d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]))
PY
done
