#!/bin/bash
# round 6: four-wave GEMM with the accumulators as physical AGPRs (invisible to the compiler) against the round-5 form ("+a"-bound C++ values; variant lib)
cd /root/repo
for rep in 1 2 3; do
  echo "--- round 5 form"; K5_LIB=kandinsky-5_amd/lib/variants/libk5_gelu26.so python tools/gemm_block_shapes.py --tokens 47616,11904,5952,3328 --kernels auto --iters 20 2>/dev/null | grep -v "^$\|^#"
  echo "--- physical AGPRs"; python tools/gemm_block_shapes.py --tokens 47616,11904,5952,3328 --kernels auto --iters 20 2>/dev/null | grep -v "^$\|^#"
done
