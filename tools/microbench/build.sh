#!/bin/bash
# builds every microbenchmark for gfx950 next to its source
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value "$f" -o "${f%.hip}" || exit 1
done
