#!/bin/bash
# interleaved A/B of the regular library against variant builds on the model's GEMM shapes: tools/gemm_variant_ab.sh name [name ...]   (lib/variants/libk5_<name>.so)
R=$PWD
for round in 1 2 3; do
  echo "base"; python $R/tools/gemm_w4_trace.py --times 2>/dev/null | python -c "import sys,json; print(' '.join(f\"{json.loads(l)['ms']*1e3:.1f}\" for l in sys.stdin))"
  for v in "$@"; do echo "$v"; K5_LIB=$R/kandinsky-5_amd/lib/variants/libk5_$v.so python $R/tools/gemm_w4_trace.py --times 2>/dev/null | python -c "import sys,json; print(' '.join(f\"{json.loads(l)['ms']*1e3:.1f}\" for l in sys.stdin))"; done
done
