#!/bin/bash
# libk5's GEMM against the vendor library on the model's shapes, same box, alternating, NO monitor kernel (a persistent 256-workgroup launch loses a
# CU to it and runs two rounds).  libk5 twice: with the epilogue the model uses, and with the plain bf16 store the vendor call does.   tools/vendor_ab.sh
R=$PWD
fmt='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l); print("libk5    %-24s %8.3f ms %8.1f TFLOP/s" % (j["shape"], j["ms"], j["tflops"]))'
for round in 1 2; do
  python $R/tools/blaslt_ref.py 2>/dev/null | grep -E "qk|out|ff1|ff2"
  python $R/tools/gemm_w4_trace.py --times 2>/dev/null | python -c "$fmt"
  python $R/tools/gemm_w4_trace.py --times --plain 2>/dev/null | python -c "$fmt"
done
