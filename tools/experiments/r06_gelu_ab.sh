#!/bin/bash
# round 6: GELU epilogue on A&S 7.1.28 (one transcendental) against 7.1.26 (two; default build of that commit; the variant lib was built with -DK5_GELU_AS26, today the switch is -DK5_GELU_AS28 for the new form), same box, alternating
cd /root/repo
for rep in 1 2 3; do
  echo "--- 7.1.26 (round 5)"; K5_LIB=kandinsky-5_amd/lib/variants/libk5_gelu26.so python tools/gemm_block_shapes.py --tokens 47616,5952,3328 --kernels auto --iters 20 2>/dev/null | grep -v "^$"
  echo "--- 7.1.28 (round 6)"; python tools/gemm_block_shapes.py --tokens 47616,5952,3328 --kernels auto --iters 20 2>/dev/null | grep -v "^$"
done
