#!/bin/bash
# tools/variant_all.sh NAME [-Dflags...] : build variants/libhz_NAME.so with EVERY .hip recompiled with the given flags
set -e
cd "$(dirname "$0")/../circuits_amd/csrc"
name=$1; shift
mkdir -p ../../variants /tmp/var_all_$name
pids=""
for src in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed -mllvm -pragma-unroll-threshold=1000000 -Wno-unused-function "$@" -c $src -o /tmp/var_all_$name/${src%.hip}.o 2>/tmp/var_all_$name/${src%.hip}.log &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
grep -l "error" /tmp/var_all_$name/*.log 2>/dev/null | head -3 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_all_$name/*.o -o ../../variants/libhz_$name.so
ls -la ../../variants/libhz_$name.so | awk '{print $5, $9}'
