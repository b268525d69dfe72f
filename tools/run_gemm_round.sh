#!/bin/bash
# GEMM regression + timing round on the GPU box:  bash tools/run_gemm_round.sh TAG
cd /root/repo; mkdir -p gpurun_out; T=${1:-x}
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -4 > gpurun_out/r05_gemm_tests_$T.log; cat gpurun_out/r05_gemm_tests_$T.log
timeout 300 python tools/gemm_fuzz.py 30 5 2>&1 | tail -3
{ timeout 300 python tools/gemm_block_shapes.py --tokens 47616,23808,11904,5952,3328 --kernels auto 2>&1 | grep -v "^#" | sed "s/^ auto/ new /"
  timeout 300 python tools/gemm_block_shapes.py --tokens 47616,23808,11904 --kernels auto --env K5_GEMM_TAIL_V2=1 2>&1 | grep -v "^#" | sed "s/^ auto/quadT/"; } > gpurun_out/r05_gemm_block_shapes_$T.log
cat gpurun_out/r05_gemm_block_shapes_$T.log
