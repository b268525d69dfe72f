cd /root/repo
for wl in 2s_256 5s_sft; do for v in 0 1; do
python bench.py --workload $wl --steps $([ $wl = 2s_256 ] && echo 20 || echo 4) --warmup $([ $wl = 2s_256 ] && echo 5 || echo 1) --no-cpu-baseline --no-live-traffic --no-vae --no-parity-check --engine-option attn_fuse_qnorm_auto=$v 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$wl auto=$v', round(d['ms_per_step'],2), {k:round(x,2) for k,x in d['kernel_time_ms_per_step'].items()}, 'used', d.get('attn_fuse_qnorm_used'), (d.get('latent_pin') or {}).get('status'), (d.get('latent_pin') or {}).get('rel_l2_update_vs_pinned'))
"
done; done
