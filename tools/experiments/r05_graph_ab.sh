cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for i in 1 2; do
echo -n "config1 eager pl0: "; one --workload 2s_256 --steps 40 --warmup 5 --no-cpu-baseline --no-vae --profile-level 0 --no-breakdown
echo -n "config1 graph pl0: "; one --workload 2s_256 --steps 40 --warmup 5 --no-cpu-baseline --no-vae --profile-level 0 --no-breakdown --graph
done
echo -n "P=8 shard eager pl0: "; one --steps 6 --warmup 2 --no-cpu-baseline --no-vae --profile-level 0 --no-breakdown --emulate-shard 8
echo -n "P=8 shard graph pl0: "; one --steps 6 --warmup 2 --no-cpu-baseline --no-vae --profile-level 0 --no-breakdown --emulate-shard 8 --graph
