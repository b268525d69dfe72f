#!/bin/bash
# full GPU regression + the bench lines of a round:  bash tools/run_full_round.sh TAG
cd /root/repo; mkdir -p gpurun_out; T=${1:-x}
timeout 1500 python -m pytest tests/ -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/r05_gputests_$T.log; tail -22 gpurun_out/r05_gputests_$T.log
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae > gpurun_out/r05_bench_$T.json 2> gpurun_out/r05_bench_$T.err
for P in 2 4 8; do timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --emulate-shard $P 2>/dev/null | grep "^{"; done > gpurun_out/r05_shards_$T.jsonl
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --workload 2s_256 2>/dev/null | grep "^{" > gpurun_out/r05_config1_$T.json
TAG=$T python - <<'P'
import json,glob,os
T=os.environ.get("T","x")
def show(path):
    for l in open(path):
        if l.startswith("{"):
            d=json.loads(l); print(os.path.basename(path), d["config"].get("emulated_shard",""), round(d["ms_per_step"],2), "attn", round(d["roofline"].get("frac",0),4) if d.get("roofline") else None, "gemm", round(d["roofline_gemm"]["frac"],4) if d.get("roofline_gemm") else None, {k:round(v,2) for k,v in d["kernel_time_ms_per_step"].items()}, d.get("parity_check",{}) and d["parity_check"].get("status"), d.get("latent_pin",{}) and d["latent_pin"].get("status"), d.get("INVALID_AS_BENCH"))
for p in sorted(glob.glob("gpurun_out/r05_*_%s.json*" % os.environ.get("TAG","x"))): show(p)
P
