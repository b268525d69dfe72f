#!/bin/bash
# round 6: ln_kernel with the per-column vectors staged once per workgroup + a row walk (K5_LN_WG workgroups) against the round-5 form
cd /root/repo
for rep in 1 2; do
echo "--- r5 kernel (lib variant)"; K5_LIB=kandinsky-5_amd/lib/variants/libk5_ln_r5.so python tools/kbench.py elem 2>/dev/null | grep ln_modulate
for wg in 512 1024 2048 4096 1000000; do echo "--- r6 kernel, K5_LN_WG=$wg"; K5_LN_WG=$wg python tools/kbench.py elem 2>/dev/null | grep ln_modulate; done
done
