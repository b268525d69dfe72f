#!/bin/bash
# round 6: k5_sample's per-call choice of where the queries are normalised ("attn_fuse_qnorm_auto"): the new test, the tests around it, then the bench line on / off alternating
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dit.py tests/test_gpu_loopback.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^\[W\|amdgpu.ids\|^$" | tail -8
for rep in 1 2; do for v in 0 1; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-parity-check --engine-option attn_fuse_qnorm_auto=$v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('attn_fuse_qnorm_auto=$v', round(d['ms_per_step'],2), 'gemm', round(d['kernel_time_ms_per_step']['gemm'],2), 'attn', round(d['kernel_time_ms_per_step']['attn_self'],2), 'elem', round(d['kernel_time_ms_per_step']['elementwise'],2), 'pin', d['latent_pin']['status'], d['latent_pin'].get('rel_l2_update_vs_pinned'), 'used', d.get('attn_fuse_qnorm_used'))
"
done; done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --no-vae --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('10 steps, default', round(d['ms_per_step'],2), 'used', d.get('attn_fuse_qnorm_used'), d['latent_pin'])
"
