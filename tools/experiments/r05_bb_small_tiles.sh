# barrier position of the 192- / 128-row forms (W4_BB6 / W4_BB4; 40 / 16 since they were written) re-swept with reads every MFMA
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "=== 40 16 (default)"; python tools/gemm_block_shapes.py --tokens 11904,5952,3328 --kernels auto --iters 30 | tail -3
for n in 48_24 56_32 64_36; do echo "=== $n"; K5_LIB=$PWD/kandinsky-5_amd/lib/variants/libk5_bb_$n.so python tools/gemm_block_shapes.py --tokens 11904,5952,3328 --kernels auto --iters 30 | tail -3; done
done
