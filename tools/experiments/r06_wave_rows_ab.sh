#!/bin/bash
# round 6, final build: 64-row attention waves (K5_ATTN_WAVE_ROWS=64; since this round also for the fused-query-norm instantiation) against the 32-row default
cd /root/repo
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-parity-check --no-breakdown "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], round(d['ms_per_step'],2), 'attn frac', round(d['roofline']['frac'],4), 'pin', (d.get('latent_pin') or {}).get('rel_l2_update_vs_pinned'), 'fused', d.get('attn_fuse_qnorm_used'))
" "$TAG"; }
for rep in 1 2 3; do
TAG="32-row waves              " run
TAG="64-row waves              " K5_ATTN_WAVE_ROWS=64 run
TAG="32-row waves, choice off  " run --engine-option attn_fuse_qnorm_auto=0
TAG="64-row waves, choice off  " K5_ATTN_WAVE_ROWS=64 run --engine-option attn_fuse_qnorm_auto=0
done
K5_ATTN_WAVE_ROWS=64 timeout 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_softmax_variants.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^\[W\|amdgpu.ids\|^$" | tail -4
