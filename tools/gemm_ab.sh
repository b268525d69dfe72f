#!/bin/bash
# A/B of GEMM kernel variants on the model's shapes (run on the GPU box): tools/gemm_ab.sh [variant_lib ...]
# each shape: baseline libk5.so, then every variant given (K5_LIB), interleaved 2 rounds; BLASLT=1 adds the vendor number once
R=$PWD
shapes="47616 3584 1792;47616 1792 1792;47616 7168 1792;47616 1792 7168;4096 4096 32768;5952 1792 7168;5952 7168 1792"
IFS=';'
for s in $shapes; do
  IFS=' ' read M N K <<< "$s"
  for round in 1 2; do
    python $R/tools/gemm_time.py $M $N $K 10 2>/dev/null | sed "s/^/base  /"
    for v in "$@"; do K5_LIB=$R/$v python $R/tools/gemm_time.py $M $N $K 10 2>/dev/null | sed "s|^|$(basename $v .so | sed s/libk5_//)  |"; done
  done
  [ -n "$BLASLT" ] && BLASLT=1 python $R/tools/gemm_time.py $M $N $K 10 2>/dev/null | grep hipBLASLt
  IFS=';'
done
