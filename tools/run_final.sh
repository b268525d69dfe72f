cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fulldepth.py -q -s -k "nabla_in_full" 2>&1 | grep -v amdgpu.ids | grep -E "NABLA in full|map entries|step |passed|failed" > gpurun_out/r05_nabla_fulldepth.log; tail -4 gpurun_out/r05_nabla_fulldepth.log
for u in 4 8 4 8; do echo "K5_GN_UNROLL=$u: $(K5_GN_UNROLL=$u timeout 300 python tools/vae_bench.py 2>&1 | tail -2 | tr '\n' ' ')"; done > gpurun_out/r05_vae_gn_unroll_ab.log; cat gpurun_out/r05_vae_gn_unroll_ab.log
timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; tail -c 1500 gpurun_out/r05_bench.json
