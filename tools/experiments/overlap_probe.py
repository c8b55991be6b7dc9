"""Does an HBM-store-bound kernel overlap a VALU-bound one on MI355X? Poseidon digest-only (integer-VALU bound) on stream A,
hipMemset-style fills (store bound) on stream B: alone and together. Prints milliseconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from circuits_amd import lib
L = lib()
n = 1 << 20
t = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x = torch.randint(0, 2**31 - 1, (n * (t - 1), 8), dtype=torch.int32)
x[:, 7] &= 0x0FFFFFFF
d_in = x.cuda(); d_out = torch.empty((n, 8), dtype=torch.int32, device="cuda")
buf = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
REP_A, REP_B = 12, 8

def run_a():
    for _ in range(REP_A):
        L.poseidon_batch_dev(t, n, d_in.data_ptr(), d_out.data_ptr(), None, sa.cuda_stream)

def run_b():
    with torch.cuda.stream(sb):
        for _ in range(REP_B):
            buf.zero_()

def timed(fns):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in fns: f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

for f in (run_a, run_b): timed([f])
a = min(timed([run_a]) for _ in range(3)); b = min(timed([run_b]) for _ in range(3)); ab = min(timed([run_a, run_b]) for _ in range(3))
print("poseidon t=%d digest x%d alone %.2f ms | fill 8 GiB x%d alone %.2f ms (%.0f GB/s) | together %.2f ms (sum %.2f, max %.2f)" % (t, REP_A, a, REP_B, b, REP_B * 8.59 / b * 1e3, ab, a + b, max(a, b)))
