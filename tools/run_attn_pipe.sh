cd /root/repo
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "64_row_waves" 2>&1 | tail -3
for r in 32 132 164 32 132; do echo "rows=$r: $(K5_ATTN_WAVE_ROWS=$r BOUNDED=1 PRESCALED=1 TIME=1 timeout 120 python tools/attn_only.py 3 2>&1 | grep attention)"; done
