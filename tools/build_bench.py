#!/usr/bin/env python3
"""Batch builder: eager host hashing vs the device DAG. Builds the benchmark's synthetic batch both ways,
checks that the circuit inputs are identical and prints where the time goes.
usage: python tools/build_bench.py [nTx nLevels maxL1Tx maxFeeTx]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import torch  # noqa: F401  (one HIP runtime per process)
except Exception:
    pass
from circuits_amd import builder as B  # noqa: E402

class Timed:
    """accumulates the time spent inside one method"""

    def __init__(self, obj, name):
        self.t, self.n = 0.0, 0
        f = getattr(obj, name)

        def g(*a, **k):
            t0 = time.time()
            try:
                return f(*a, **k)
            finally:
                self.t += time.time() - t0
                self.n += 1
        setattr(obj, name, g)


shape = tuple(int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 32, 256, 64)
B.synthetic_batch(8, 16, 2, 2, device=0)   # warm up: library load, kernel code objects
sign = Timed(B.Account, "sign_msg")
t0 = time.time()
dev = B.synthetic_batch(*shape, device=0)
t_dev = time.time() - t0
t_sign_dev = sign.t
st = dict(dev.db.hasher.stats)
hostp = Timed(B.host(), "poseidon")
t0 = time.time()
eager = B.synthetic_batch(*shape)
t_eager = time.time() - t0
same = eager.get_input() == dev.get_input() and eager.get_hash_inputs() == dev.get_hash_inputs()
n_sig = sum(1 for t in dev.txs if not t.get("onChain"))
print(json.dumps({"shape": shape, "identical_inputs": same, "eager_host_s": round(t_eager, 3), "device_dag_s": round(t_dev, 3),
                  "dag_jobs": st["jobs"], "dag_launches": st["segments"], "dag_device_ms": round(st["device_ms"], 3),
                  "dag_resolve_s_incl_copies_and_python": round(st["resolve_s"], 3), "signatures_signed_on_host": n_sig, "signing_s": round(t_sign_dev, 3),
                  "eager_host_poseidon_calls": hostp.n, "eager_host_poseidon_s": round(hostp.t, 3)}))
