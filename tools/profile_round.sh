#!/bin/bash
# Round profile: everything the numbers in bench.py / DESIGN.md are checked against.  Run on the GPU box from the repo root:
#     tools/profile_round.sh r01
# writes gpurun_out/prof_<tag>/... (scratch) and the summaries gpurun_out/<tag>_*.{md,json} to copy into profiles/.
TAG=${1:-r01}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp
if [ -z "$LINES_ONLY" ]; then   # LINES_ONLY=1: only the workload / shard lines of steps 6-8 (a rebuild that does not touch the profiled default path)
# 1. kernel trace + stats of the bench command (no counters in this pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-breakdown > $OUT/${TAG}_bench_under_rocprof.log 2>&1
grep "^{" $OUT/${TAG}_bench_under_rocprof.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json   # the line of THIS run: what the kernel-trace figure must agree with
# 2. HBM traffic counters, one per pass (2 visual blocks are enough: per-launch numbers)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-breakdown --blocks 2 > /dev/null 2>&1
done
# 3. issue / wait counters of the attention kernel, launched the way the engine launches it (fixed softmax offset, pre-scaled keys)
export BOUNDED=1 PRESCALED=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_sq -o p -- python $R/tools/attn_only.py 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_lds -o p -- python $R/tools/attn_only.py 2 > /dev/null 2>&1
# 4. VAE decode kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_vae -o vae -- python $R/tools/vae_bench.py > $OUT/${TAG}_vae_under_rocprof.log 2>&1
# 5. NABLA (10 s, 93 696 tokens) kernel stats: the map kernels next to the sparse attention
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_nabla -o nabla -- python $R/bench.py --workload 10s_nabla --steps 1 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-breakdown > $OUT/${TAG}_nabla_under_rocprof.log 2>&1
cd $R
python tools/profile_summarize.py $TAG
fi
cd $R
# 6. the other workloads and the emulated shard sizes, one JSON line each (no profiler)
: > $OUT/${TAG}_workloads.jsonl; : > $OUT/${TAG}_shards.jsonl
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --attn-online 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --qk-gain 3 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --qk-gain 6 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl   # beyond the window: anchored offsets
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --qk-gain 6 --engine-option attn_anchor=0 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl   # ... the online form
for m in 1 3 7; do python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --fp8 $m 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl; done   # W8A8 e4m3: FF | FF + q,k,v | + out (lossy, INVALID_AS_BENCH)
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --workload 5s_sft 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-vae --workload 2s_256 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl
python bench.py --steps 45 --warmup 5 --no-cpu-baseline --no-live-traffic --no-vae --magcache 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl
for np in 0.9 0.15 0.0; do python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --workload 10s_nabla --nabla-p $np 2>/dev/null | grep "^{" >> $OUT/${TAG}_workloads.jsonl; done
for sh in 2 4 8; do for sl in 1 2; do python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --emulate-shard $sh --sp-slices $sl 2>/dev/null | grep "^{" >> $OUT/${TAG}_shards.jsonl; done; done
# 7. NABLA under sequence parallelism: the 10 s clip as rank 0 of 4 (BASELINE config 4), near-dense and STA-only maps, one / two passes
for np in 0.9 0.0; do for ps in 1 2; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --workload 10s_nabla --nabla-p $np --emulate-shard 4 --engine-option sp_nabla_passes=$ps 2>/dev/null | grep "^{" >> $OUT/${TAG}_shards.jsonl; done; done
# 8. BASELINE config 5's shape (1280x768, 10 s: 3660 blocks) as rank 0 of 4, STA-only and near-dense maps
for np in 0.0 0.9; do python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --workload 10s_hd_nabla --nabla-p $np --emulate-shard 4 2>/dev/null | grep "^{" >> $OUT/${TAG}_shards.jsonl; done
python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --workload 10s_hd_nabla --nabla-p 0.0 --emulate-shard 4 --fp8 3 2>/dev/null | grep "^{" >> $OUT/${TAG}_shards.jsonl   # BASELINE config 5 is the fp8 configuration
wc -l $OUT/${TAG}_workloads.jsonl $OUT/${TAG}_shards.jsonl
