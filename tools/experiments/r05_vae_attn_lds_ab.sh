# mid-block attention of the VAE: K-row stride 1056 + V^T swizzle (0,2,3,1) against rounds 3-5 (1040, plain quarter index): tests, conflicts, time
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vae.py -x -q -m gpu 2>&1 | tail -2
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcl_va_new -o p -- python $R/tools/vae_bench.py > $R/gpurun_out/va_new.log 2>&1
K5_LIB=$R/kandinsky-5_amd/lib/variants/libk5_va_old.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcl_va_old -o p -- python $R/tools/vae_bench.py > $R/gpurun_out/va_old.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for tag in ("new", "old"):
    acc = collections.defaultdict(float); dur = []
    for fn in glob.glob(f"gpurun_out/pmcl_va_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "vae_attn512" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    for fn in glob.glob(f"gpurun_out/pmcl_va_{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "vae_attn512" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(tag, dict(acc), "vae_attn512 launches", len(dur), "avg us (under counters)", sum(dur) / max(len(dur), 1))
PY
for i in 1 2; do
echo -n "new: "; python tools/vae_bench.py 2>&1 | tail -1
echo -n "old: "; K5_LIB=$PWD/kandinsky-5_amd/lib/variants/libk5_va_old.so python tools/vae_bench.py 2>&1 | tail -1
done
