#!/bin/bash
# round 6: "attn_fuse_qnorm" (norm_qk + RoPE of the visual queries inside the attention kernel's Q load) on the final build, on / off alternating on one box
cd /root/repo
for rep in 1 2 3; do for v in 0 1; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-parity-check --engine-option attn_fuse_qnorm=$v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('attn_fuse_qnorm=$v', round(d['ms_per_step'],2), 'gemm', round(d['kernel_time_ms_per_step']['gemm'],2), 'attn', round(d['kernel_time_ms_per_step']['attn_self'],2), 'elem', round(d['kernel_time_ms_per_step']['elementwise'],2), d['latent_pin']['status'] if d.get('latent_pin') else None)
"
done; done
