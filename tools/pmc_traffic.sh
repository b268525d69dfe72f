#!/bin/bash
# HBM traffic of the dominant kernels (guide: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes; on gfx950 FETCH_SIZE
# reports half of a wide coalesced stream -> doubled below; units: KB).
export TMPDIR=/tmp; R=$PWD; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --blocks 2 > /dev/null 2>&1
done
cd $R
python - <<PY
import csv, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(f"gpurun_out/pmc_{c}/p_counter_collection.csv")):
        if r["Counter_Name"] == c:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for d, v in per.items():
        res[names[d]][c].append(v)
out = {}
for k, d in res.items():
    if not any(t in k for t in ("attn_fwd", "gemm_bf16", "ln_kernel", "rmsnorm")): continue
    f = max(d["FETCH_SIZE"]) if d["FETCH_SIZE"] else 0; w = max(d["WRITE_SIZE"]) if d["WRITE_SIZE"] else 0
    out[k[:70]] = {"fetch_MB_raw(max launch)": f / 1024, "fetch_MB_x2_gfx950": 2 * f / 1024, "write_MB": w / 1024, "launches": len(d["FETCH_SIZE"])}
print(json.dumps(out, indent=1))
PY
