"""Quick device timing of the Poseidon batch kernel (digest and witness modes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from circuits_amd import lib
L = lib()
dev = torch.device("cuda:0")
for t in (3, 5):
    n = 1 << 20
    nsbox = 8 * t + [56, 57, 56, 60, 60, 63][t - 2]
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randint(0, 2**31 - 1, (n * (t - 1), 8), dtype=torch.int32, generator=g)
    x[:, 7] &= 0x0FFFFFFF  # < 2^252 < r
    d_in = x.to(dev)
    d_out = torch.empty((n, 8), dtype=torch.int32, device=dev)
    d_wit = torch.empty((3 * nsbox * n, 8), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for mode, wit in (("digest", None), ("witness", d_wit.data_ptr())):
        L.poseidon_batch_dev(t, n, d_in.data_ptr(), d_out.data_ptr(), wit, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            L.poseidon_batch_dev(t, n, d_in.data_ptr(), d_out.data_ptr(), wit, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        by = n * (32 * (t - 1) + 32 + (96 * nsbox if wit else 0))
        print("t=%d %s: %.3f ms  %.1f Mperm/s  %.1f GB/s" % (t, mode, ms, n / ms / 1e3, by / ms / 1e6))
