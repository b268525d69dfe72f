# 256-row form: all 16 DMA pieces of a K-tile issued BEFORE the fragment-read window (a piece costs more to issue next to ds_reads), W4P = DMA spacing, barrier, read spacing
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "=== default (8,80,1)"; python tools/gemm_block_shapes.py --tokens 47616 --kernels auto --iters 20 | tail -1
for n in 5_80_1 5_88_1 6_88_1 6_96_1 4_64_1; do echo "=== $n"; K5_LIB=$PWD/kandinsky-5_amd/lib/variants/libk5_w4p_$n.so python tools/gemm_block_shapes.py --tokens 47616 --kernels auto --iters 20 | tail -1; done
done
