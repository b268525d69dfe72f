#!/bin/bash
# round 6: the two norm passes (ln_kernel with LDS-staged vectors + burst loads; rmsnorm_rope with the next row prefetched) against the round-5 kernels
cd /root/repo
for rep in 1 2 3; do
echo "--- r5 kernels (lib variant)"; K5_LIB=kandinsky-5_amd/lib/variants/libk5_ln_r5.so python tools/kbench.py elem 2>/dev/null | grep "ln_modulate\|rmsnorm"
echo "--- r6 kernels"; python tools/kbench.py elem 2>/dev/null | grep "ln_modulate\|rmsnorm"
done
