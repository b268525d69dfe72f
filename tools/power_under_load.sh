#!/bin/bash
# What the board draws and which clock the firmware grants while the hot kernels run (rocm-smi samples every 0.5 s next to the loop).
#   tools/power_under_load.sh           -> gpurun_out/r04_power_under_load.log
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; L=$OUT/r04_power_under_load.log; : > $L
sample() {  # $1 = label, $2 = seconds
  for i in $(seq 1 $((2 * $2))); do
    echo "$1 | $(rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E 'Power|sclk|fclk|mclk|Max Graphics' | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ';')" >> $L
    sleep 0.5
  done
}
echo "# idle" >> $L; sample idle 2
for case in "attention:BOUNDED=1 PRESCALED=1 python tools/attn_only.py 600" "gemm 47616x1792x7168:python tools/gemm_only.py 47616 1792 7168 8000" "gemm 47616x7168x1792:python tools/gemm_only.py 47616 7168 1792 7000"; do
  name=${case%%:*}; cmd=${case#*:}
  echo "# $name" >> $L
  ( eval "timeout 60 env $cmd" > /dev/null 2>&1 ) &
  pid=$!
  sleep 4          # import + warm-up
  sample "$name" 4
  wait $pid
done
cat $L
