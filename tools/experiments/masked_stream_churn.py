"""How many HZ_FLAG_LATENCY contexts can a process create, run and destroy? (every CU-masked stream owns a hardware queue; kernels
with scratch take their scratch per queue.) The sequence of bench.py's sweep at the headline shape: plain contexts, then 2 and 4
flagged ones in flight, several rounds.
python tools/experiments/masked_stream_churn.py [rounds] [big: 0 | 1]"""
import os
import sys

# (round 5 set HZ_MAX_PARTITIONED=4 here against the library's default of two; the default is four since round 6)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circuits_amd import lib, builder as B   # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
big = len(sys.argv) > 2 and sys.argv[2] == "1"
L = lib()
shape = (2048, 32, 256, 64) if big else (64, 16, 8, 4)
bb = B.synthetic_batch(*shape, n_accounts=2048 if big else 32, exits=3, seed=77)
inp = bb.get_input()
streams = [torch.cuda.Stream() for _ in range(4)]
made = 0
for r in range(rounds):
    for per, flags in ((2, 0), (2, 2), (4, 2), (3, 2)):
        cs = [L.ctx("rollup-main", nTx=shape[0], nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3], flags=flags) for _ in range(per)]
        cs[0].set_inputs(inp)
        for c in cs[1:]:
            c.set_inputs(inp)
        for it in range(6):
            for k, c in enumerate(cs):
                c.enqueue(streams[k].cuda_stream)
            for c in cs:
                c.check()
        assert cs[-1].get("main.hashGlobalInputs") == bb.get_hash_inputs()
        made += per
        del cs, c
        print("round %d: %d contexts with flags %d ran; %d created so far" % (r, per, flags, made), flush=True)
print("ok")
