#!/bin/bash
# round 6 (review item "sparse attention: persistent workgroups, or close the book with a measurement"): what does a sparse JOB cost beyond its tiles?
# The 10 s 768x512 clip (93 696 tokens = 1464 blocks, 28 heads), 128-query workgroups (20 496 jobs per launch whatever the map), the map's
# density swept through NABLA's P: time per launch = jobs x c + executed tiles x tau.  c x jobs / time at the operating density is the most a
# persistent-workgroup form (next job's first DMAs issued ahead of this job's epilogue) could win.  Fit: tools/experiments/r06_sparse_job_cost_fit.py
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/sparse_job_cost.jsonl
for g in 2 4; do
for p in 0.0 0.02 0.05 0.1 0.15 0.25 0.4; do
  python bench.py --workload 10s_nabla --nabla-p $p --engine-option nabla_group_rows=$g --blocks 4 --steps 3 --warmup 1 --no-vae --no-cpu-baseline \
      --no-live-traffic --no-parity-check --no-breakdown 2>/dev/null | tail -1 >> gpurun_out/sparse_job_cost.jsonl
done
done
python tools/experiments/r06_sparse_job_cost_fit.py gpurun_out/sparse_job_cost.jsonl | tee gpurun_out/sparse_job_cost_fit.txt
