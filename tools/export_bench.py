"""Device export of the witness in variable order, timed at the headline shape (RollupMain(2048, 32, 256, 64), B batches resident):
per-instance export in component-major order (the shape of a reducing circom compile's numbering) and in a random permutation,
HIP events on the export's own stream. python tools/export_bench.py [B] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from circuits_amd import lib, builder as B   # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    L = lib()
    shape = (2048, 32, 256, 64)
    g = L.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], n_instances=nb)
    bb = B.synthetic_batch(*shape, n_accounts=2048, exits=32, seed=0x48455A31)
    g.set_inputs(bb.get_input(), instance=0)
    for k in range(1, nb):
        g.copy_instance_inputs(0, k)
    g.run()
    assert g.get("main.hashGlobalInputs") == bb.get_hash_inputs()
    wl = g.witness_len()
    out = torch.zeros(wl * 32, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.Stream()
    cm = g.component_major_index()
    perm = np.concatenate([[0], 1 + np.random.default_rng(11).permutation(wl - 1)]).astype(np.uint64)
    ident = np.arange(wl, dtype=np.uint64)
    for name, index in (("component-major", cm), ("own-order", ident), ("permuted", perm)):
        mp = g.symmap_from_index(index)
        t0 = time.time()
        tab = mp.upload()
        t_plan = time.time() - t0
        times = []
        for r in range(reps + 1):
            inst = r % nb
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record(s)
            mp.export_dev(out.data_ptr(), inst, stream=s.cuda_stream)
            ev1.record(s)
            torch.cuda.synchronize()
            if r:
                times.append(ev0.elapsed_time(ev1))
        ms = sorted(times)[len(times) // 2]
        print("export %-16s B=%d: %.3f ms per batch (min %.3f)  %.2f TB/s read+write   plan %.1f s  tables %.0f MB" % (
            name, nb, ms, min(times), 2 * wl * 32 / ms / 1e9, t_plan, tab / 1e6), flush=True)
        if name == "component-major":   # several batches per call: the sections whose unit is the instance are read as whole lines
            for cnt in (4, 8):
                if cnt > nb:
                    continue
                big = torch.zeros(cnt * wl * 32, dtype=torch.uint8, device="cuda:0")
                tt = []
                for r in range(reps + 1):
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    ev0.record(s)
                    L._check(L.c.hz_witness_export_range_dev(g.h, mp.h, (r * cnt) % (nb - cnt + 1) // 4 * 4, cnt, big.data_ptr(), s.cuda_stream))
                    ev1.record(s)
                    torch.cuda.synchronize()
                    if r:
                        tt.append(ev0.elapsed_time(ev1) / cnt)
                print("export %-16s B=%d, %d batches per call: %.3f ms per batch  %.2f TB/s read+write" % (name, nb, cnt, sorted(tt)[len(tt) // 2], 2 * wl * 32 / sorted(tt)[len(tt) // 2] / 1e9), flush=True)
                del big
        del mp


if __name__ == "__main__":
    main()
