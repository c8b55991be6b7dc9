#!/bin/bash
# tools/variant.sh NAME FILE.hip [-Dflags...] : build gpurun_out/libhz_NAME.so with FILE.hip recompiled with the given flags
set -e
cd "$(dirname "$0")/../circuits_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-function "$@" -c $src -o /tmp/var_$name.o 2>&1 | grep -E "error" | head -5 || true
objs=""
for o in build/*.o; do [ "$o" = "build/${src%.hip}.o" ] && objs="$objs /tmp/var_$name.o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../variants/libhz_$name.so 2>&1 | grep -E "error" | head -5 || true
ls -la ../../variants/libhz_$name.so | awk '{print $5, $9}'
