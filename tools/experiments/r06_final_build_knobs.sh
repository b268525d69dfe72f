#!/bin/bash
# round 6, final build: the bit-identical knobs that were "neutral" on earlier builds, once more (alternating): 64-row attention waves, hipGraph replay of the step
cd /root/repo
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-vae --no-parity-check --no-breakdown "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], round(d['ms_per_step'],2), 'attn frac', round(d['roofline']['frac'],4), (d.get('latent_pin') or {}).get('rel_l2_update_vs_pinned'))
" "$TAG"; }
for rep in 1 2; do
TAG="default        " run
TAG="wave rows 64   " K5_ATTN_WAVE_ROWS=64 run
TAG="hipGraph       " run --graph --profile-level 0
done
