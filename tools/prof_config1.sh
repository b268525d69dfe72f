#!/bin/bash
# kernel stats of BASELINE config 1 (3328 tokens): which launches the small-N step is made of.   bash tools/prof_config1.sh TAG
TAG=${1:-r05}; export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_c1 -o c1 -- python $R/bench.py --workload 2s_256 --steps 8 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --no-breakdown > $OUT/${TAG}_c1_under_rocprof.log 2>&1
cd $R; python tools/rocprof_summary.py gpurun_out/prof_${TAG}_c1 gpurun_out/${TAG}_config1_kernel_stats.md "BASELINE config 1 (2s 256x256, 3328 tokens), 10 steps x 32 blocks" | head -45
