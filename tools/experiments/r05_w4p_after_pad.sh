# the 256-row form's schedule knobs (DMA spacing, barrier position, fragment-read spacing) re-swept now that a fragment read takes 4 LDS cycles, not 8
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "=== default (8,64,2)"; python tools/gemm_block_shapes.py --tokens 47616 --kernels auto --iters 20 | tail -1
for n in 8_64_1 8_80_1 8_88_1 8_96_1; do echo "=== $n"; K5_LIB=$PWD/kandinsky-5_amd/lib/variants/libk5_w4p_$n.so python tools/gemm_block_shapes.py --tokens 47616 --kernels auto --iters 20 | tail -1; done
done
