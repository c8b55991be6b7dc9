#!/bin/bash
# Build the trees tools/experiments/regression_ab.sh compares, under variants/ab/<name>/ (git-ignored; they travel to the GPU box).
cd "$(git rev-parse --show-toplevel)"
for pair in r4:e8a27d7 a_e98be96:e98be96 b_bc0ddb2:bc0ddb2 c_7d8a146:7d8a146 d_67cadea:67cadea e_07fcb01:07fcb01 head5:23aaee9; do
  n=${pair%%:*}; c=${pair##*:}; mkdir -p variants/ab/$n
  git archive $c circuits_amd include bench.py tools | tar -x -C variants/ab/$n
  make -j${JOBS:-4} -C variants/ab/$n/circuits_amd/csrc ../libhermez_witness.so ../libhz_host.so > variants/ab/$n/build.log 2>&1 && echo "$n ok" || echo "$n FAILED"
done
