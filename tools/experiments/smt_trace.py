#!/usr/bin/env python3
"""Per-wavefront timeline of the transaction launch of k_smt (HZ_SMT_TRACE): where do the 30 % between the kernel's time and its
instructions / (SIMDs x clock) go?  usage (GPU box):  HZ_SMT_TRACE=/tmp/smt.trace python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-node
--no-e2e --no-withdraw --no-poseidon --no-sweep --no-deep-state --no-export ; python tools/experiments/smt_trace.py /tmp/smt.trace"""
import struct
import sys

import numpy as np

data = open(sys.argv[1], "rb").read()
off, recs = 0, []
while off < len(data):
    n = struct.unpack_from("<Q", data, off)[0]
    recs.append(np.frombuffer(data, dtype=np.uint64, count=n // 8, offset=off + 8).reshape(-1, 4))
    off += 8 + n
for ri, r in enumerate(recs):
    r = r[r[:, 1] > 0]
    if not len(r):
        continue
    t0, t1 = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0   # microseconds (100 MHz)
    dur = e - s
    chain = ((r[:, 3] >> np.uint64(24)) & np.uint64(15)).astype(int)
    thr = (r[:, 3] >> np.uint64(32)).astype(int)
    hw = r[:, 2] & np.uint64(0xffffffff)
    xcc = (r[:, 2] >> np.uint64(32)) & np.uint64(15)
    cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
    se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int)
    simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
    print("context record %d: %d wavefronts, kernel span %.2f ms, sum of wave lifetimes %.1f ms (/ 2048 slots = %.2f ms)" % (ri, len(r), e.max() / 1e3, dur.sum() / 1e3, dur.sum() / 2048 / 1e3))
    for c in range(4):
        m = chain == c
        if m.any():
            print("  chain %d: %5d waves, lifetime mean %.2f ms (min %.2f, max %.2f), start mean %.2f ms, thr_wave mean %.1f (0 on %d waves)" % (
                c, m.sum(), dur[m].mean() / 1e3, dur[m].min() / 1e3, dur[m].max() / 1e3, s[m].mean() / 1e3, thr[m].mean(), (thr[m] == 0).sum()))
    # occupancy over time: wavefronts resident per 0.5 ms
    edges = np.arange(0, e.max() + 500, 500)
    occ = [(np.minimum(e, hi) - np.maximum(s, lo)).clip(min=0).sum() / 500 for lo, hi in zip(edges[:-1], edges[1:])]
    print("  wavefronts resident per 0.5 ms bin:", " ".join("%d" % round(x) for x in occ))
    # per (xcc, se, cu, simd) slot load
    key = xcc.astype(np.int64) * 4096 + se * 512 + cu * 16 + simd
    uniq, inv = np.unique(key, return_inverse=True)
    load = np.bincount(inv, weights=dur) / 1e3
    cnt = np.bincount(inv)
    print("  distinct (xcc, se, cu, simd): %d; waves per SIMD min %d max %d; busy ms per SIMD min %.2f mean %.2f max %.2f" % (len(uniq), cnt.min(), cnt.max(), load.min(), load.mean(), load.max()))
    last = np.zeros(len(uniq)); np.maximum.at(last, inv, e / 1e3)
    print("  last wave end per SIMD: min %.2f mean %.2f max %.2f ms" % (last.min(), last.mean(), last.max()))
