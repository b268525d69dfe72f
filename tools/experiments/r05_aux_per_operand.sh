# cache policy of the operand DMAs per operand (aux bits of buffer_load ... lds: 1 = sc0, 2 = nt, 16 = sc1), 47 616-row block
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "=== default"; python tools/gemm_block_shapes.py --tokens 47616 --kernels auto --iters 20 | tail -1
for n in W2 X2 W1 W16; do echo "=== $n"; K5_LIB=$PWD/kandinsky-5_amd/lib/variants/libk5_aux_$n.so python tools/gemm_block_shapes.py --tokens 47616 --kernels auto --iters 20 | tail -1; done
done
