# 32-B piece pad (conflict-free fragment reads under the real ds_read_b128 lane groups) against the 16-B pad of rounds 2-5, same box
cd $GRAFT_REPO_ROOT
V=$PWD/kandinsky-5_amd/lib/variants/libk5_pad1040.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
bash tools/pmc_gemm.sh r05p_ff1 47616 7168 1792 2>&1 | tail -2
for i in 1 2; do
echo "=== pad 1056"; python tools/gemm_block_shapes.py --tokens 47616,11904,5952,3328 --kernels auto --iters 20
echo "=== pad 1040"; K5_LIB=$V python tools/gemm_block_shapes.py --tokens 47616,11904,5952,3328 --kernels auto --iters 20
done
