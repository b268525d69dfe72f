# new default (32-B pad, W4P = 8, 80, 1) against the 32-B pad with the old knobs (8, 64, 2) on the small / shard shapes (the read spacing applies to every tile height) and through the engine
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
V=$PWD/kandinsky-5_amd/lib/variants/libk5_w4p_old.so
for i in 1 2; do
echo "=== new default"; python tools/gemm_block_shapes.py --tokens 47616,11904,5952,3328 --kernels auto --iters 20 | tail -4
echo "=== 8,64,2"; K5_LIB=$V python tools/gemm_block_shapes.py --tokens 47616,11904,5952,3328 --kernels auto --iters 20 | tail -4
done
one() { python bench.py "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['kernel_time_ms_per_step'].items()})"; }
P=$PWD/kandinsky-5_amd/lib/variants/libk5_pad1040.so
for i in 1 2; do
echo -n "config2 new: "; one --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-live-traffic --no-parity-check
echo -n "config2 pad1040: "; K5_LIB=$P one --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-live-traffic --no-parity-check
done
