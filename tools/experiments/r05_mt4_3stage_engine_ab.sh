# config 1 and the emulated 8-GPU shard through the engine, three-stage (default build) vs two-stage (variant) 128-row GEMM form, interleaved on one box
cd $GRAFT_REPO_ROOT
V=$PWD/kandinsky-5_amd/lib/variants/libk5_mt4_2stage.so
one() { python bench.py "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_time_ms_per_step'].items()})"; }
for i in 1 2 3; do
  echo -n "3-stage config1: "; one --workload 2s_256 --steps 30 --warmup 5 --no-cpu-baseline --no-vae
  echo -n "2-stage config1: "; K5_LIB=$V one --workload 2s_256 --steps 30 --warmup 5 --no-cpu-baseline --no-vae
done
