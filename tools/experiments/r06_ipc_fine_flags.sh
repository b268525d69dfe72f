#!/bin/bash
# round 6: the IPC transport with its flag page in fine-grained memory (csrc/ipc_comm.h Group::open): probe with both kinds, then the cross-process tests
export K5_IPC_TIMEOUT_S=60
mkdir -p gpurun_out
for kind in fine coarse; do
  if [ $kind = coarse ]; then export K5_IPC_COARSE_FLAGS=1; else unset K5_IPC_COARSE_FLAGS; fi
  timeout 100 tools/probes/ipc_probe.bin 4 64 12 > gpurun_out/r06_ipc_probe_4_$kind.log 2>&1; echo "probe rc $?" >> gpurun_out/r06_ipc_probe_4_$kind.log
  tail -4 gpurun_out/r06_ipc_probe_4_$kind.log
done
unset K5_IPC_COARSE_FLAGS
timeout 100 tools/probes/ipc_probe.bin 8 32 8 > gpurun_out/r06_ipc_probe_8_fine.log 2>&1; echo "probe rc $?" >> gpurun_out/r06_ipc_probe_8_fine.log; tail -2 gpurun_out/r06_ipc_probe_8_fine.log
timeout 2400 python -m pytest tests/test_gpu_ipc_ranks.py tests/test_gpu_cli_ranks.py tests/test_bench_contract.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r06_ipc_tests_fine.log 2>&1; echo "tests rc $?" >> gpurun_out/r06_ipc_tests_fine.log
grep -v "^\[W\|amdgpu.ids\|^$\|Gloo" gpurun_out/r06_ipc_tests_fine.log | tail -12
grep -h "ipc_flags_finegrained" -r /tmp/pytest-of-root 2>/dev/null | head -2 | cut -c1-300
