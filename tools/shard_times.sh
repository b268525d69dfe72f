#!/bin/bash
# per-rank compute time of a P-GPU sequence-parallel run, emulated on one GPU (collectives move nothing)
for P in "$@"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --emulate-shard $P 2>/dev/null | grep "^{" | P=$P python -c "
import json, sys, os
d = json.loads(sys.stdin.read())
print('P=' + os.environ['P'], round(d['ms_per_step'], 1), 'ms/step', {k: round(v, 1) for k, v in d['kernel_time_ms_per_step'].items()})"
done
