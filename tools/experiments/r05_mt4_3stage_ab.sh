set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -5
echo "=== 3-stage, MT forced 4"
K5_GEMM_MT=4 python tools/gemm_block_shapes.py --tokens 11904,5952,3328 --kernels 4 --iters 30
echo "=== 2-stage, MT forced 4"
K5_LIB=$PWD/kandinsky-5_amd/lib/variants/libk5_mt4_2stage.so K5_GEMM_MT=4 python tools/gemm_block_shapes.py --tokens 11904,5952,3328 --kernels 4 --iters 30
echo "=== MT forced 6"
K5_GEMM_MT=6 python tools/gemm_block_shapes.py --tokens 11904,5952,3328 --kernels 4 --iters 30
echo "=== MT forced 8"
K5_GEMM_MT=8 python tools/gemm_block_shapes.py --tokens 11904,5952,3328 --kernels 4 --iters 30
echo "=== auto (3-stage lib)"
python tools/gemm_block_shapes.py --tokens 11904,5952,3328 --kernels auto --iters 30
python bench.py --workload 2s_256 --steps 20 --warmup 5 --no-cpu-baseline --no-vae 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config1', d['ms_per_step'], d['kernel_time_ms_per_step'])"
