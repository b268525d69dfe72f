#!/bin/bash
# round 6: the split-K tail of the four-wave GEMM through the engine (config 2, 4 steps), on / off alternating on one box
cd /root/repo
for rep in 1 2 3; do for v in 0 1; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-parity-check --engine-option gemm_split_tail=$v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('gemm_split_tail=$v', round(d['ms_per_step'],2), 'gemm', round(d['kernel_time_ms_per_step']['gemm'],2), 'attn', round(d['kernel_time_ms_per_step']['attn_self'],2), 'elem', round(d['kernel_time_ms_per_step']['elementwise'],2))
"
done; done
for v in 0 1; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-vae --workload 2s_256 --engine-option gemm_split_tail=$v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config 1 gemm_split_tail=$v', round(d['ms_per_step'],2), {k:round(x,2) for k,x in d['kernel_time_ms_per_step'].items()})
"; done
for v in 0 1; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --workload 10s_nabla --nabla-p 0.15 --engine-option gemm_split_tail=$v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('10s nabla gemm_split_tail=$v', round(d['ms_per_step'],2), {k:round(x,2) for k,x in d['kernel_time_ms_per_step'].items()})
"; done
