#!/bin/bash
# Build an A/B variant of libk5.so: tools/build_variant.sh NAME FILE.hip [-DFLAG ...]  ->  kandinsky-5_amd/lib/variants/libk5_NAME.so
# (one source recompiled with extra flags, every other object taken from the regular build; run with K5_LIB=<that path>)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python kandinsky-5_amd/build.py > /dev/null
mkdir -p kandinsky-5_amd/lib/variants kandinsky-5_amd/build/variants
obj=kandinsky-5_amd/build/variants/${name}_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I /opt/rocm/include -I include -I kandinsky-5_amd/csrc -fno-slp-vectorize -Wno-unused-result "$@" -c kandinsky-5_amd/csrc/$src -o $obj
others=$(ls kandinsky-5_amd/build/*.o | grep -v "/$(basename ${src%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kandinsky-5_amd/lib/variants/libk5_${name}.so $others $obj -ldl
echo kandinsky-5_amd/lib/variants/libk5_${name}.so
