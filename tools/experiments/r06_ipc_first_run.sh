#!/bin/bash
# round 6: first run of the IPC transport through the engine (probe, tiny plumbing run, the cross-process tests, the bench line)
export K5_IPC_TIMEOUT_S=30
mkdir -p gpurun_out
timeout 100 tools/probes/ipc_probe.bin 4 64 12 > gpurun_out/r06_ipc_probe_4.log 2>&1; echo "probe rc $?" >> gpurun_out/r06_ipc_probe_4.log
timeout 100 tools/probes/ipc_probe.bin 8 32 8 > gpurun_out/r06_ipc_probe_8.log 2>&1; echo "probe rc $?" >> gpurun_out/r06_ipc_probe_8.log
K5_SP_TRANSPORT=ipc K5_OVERSUBSCRIBE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  tests/ipc_rank_worker.py --case c1 --tiny --out gpurun_out/ipc_tiny > gpurun_out/r06_ipc_tiny.log 2>&1; echo "tiny rc $?" >> gpurun_out/r06_ipc_tiny.log
tail -5 gpurun_out/r06_ipc_tiny.log
timeout 2400 python -m pytest tests/test_gpu_ipc_ranks.py -x -q -s -m gpu > gpurun_out/r06_ipc_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r06_ipc_tests.log
tail -30 gpurun_out/r06_ipc_tests.log
timeout 900 python bench.py --gpus 2 --oversubscribe --steps 2 --warmup 1 --no-vae --no-breakdown > gpurun_out/r06_bench_ipc2.json 2> gpurun_out/r06_bench_ipc2.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r06_bench_ipc2.json; tail -5 gpurun_out/r06_bench_ipc2.err
