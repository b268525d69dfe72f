export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/r4c; mkdir -p $OUT; cd /tmp
for mode in default TAIL_V1 NO_TAIL; do
  case $mode in default) E="";; TAIL_V1) E="K5_GEMM_TAIL_V1=1";; NO_TAIL) E="K5_GEMM_NO_TAIL=1";; esac
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$mode -o kt -- python $R/tools/gemm_w4_trace.py --times > /dev/null 2>&1
  f=$(find $OUT/kt_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; head -12 $f | cut -c1-200
done
