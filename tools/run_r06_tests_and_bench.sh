#!/bin/bash
# round 6: the whole -m gpu suite + the default bench line:  bash tools/run_r06_tests_and_bench.sh TAG
cd /root/repo; mkdir -p gpurun_out; T=${1:-x}
export K5_IPC_TIMEOUT_S=120
timeout 3000 python -m pytest tests/ -q -m gpu --durations=10 -p no:cacheprovider > gpurun_out/r06_gputests_$T.full.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputests_$T.full.log
grep -v "^\[W\|amdgpu.ids\|^$\|Gloo" gpurun_out/r06_gputests_$T.full.log | tail -30 > gpurun_out/r06_gputests_$T.log; tail -25 gpurun_out/r06_gputests_$T.log
timeout 900 python bench.py > gpurun_out/r06_bench_$T.json 2> gpurun_out/r06_bench_$T.err; echo "bench rc $?"
python - <<P
import json
for l in open("gpurun_out/r06_bench_$T.json"):
    if l.startswith("{"):
        d=json.loads(l); print(round(d["ms_per_step"],2), "attn", round(d["roofline"]["frac"],4), "gemm", round(d["roofline_gemm"]["frac"],4), {k:round(v,2) for k,v in d["kernel_time_ms_per_step"].items()}, d["parity_check"]["status"], d["latent_pin"]["status"], d.get("INVALID_AS_BENCH"), d["e2e_clip_s"])
P
