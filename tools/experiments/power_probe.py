"""Clock / power of the device while (a) an integer-VALU-bound kernel, (b) a store-bound fill, (c) both run. rocm-smi is sampled
from a thread. Usage: python power_probe.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from circuits_amd import lib
L = lib()
n = 1 << 20
t = 3
x = torch.randint(0, 2**31 - 1, (n * (t - 1), 8), dtype=torch.int32); x[:, 7] &= 0x0FFFFFFF
d_in = x.cuda(); d_out = torch.empty((n, 8), dtype=torch.int32, device="cuda")
buf = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
samples = []
stop = False

def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "-c", "-P", "--csv"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5).stdout
            samples.append((time.time(), o.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.time(), "err %s" % e))
        time.sleep(0.3)

def phase(name, a, b, secs=4.0):
    t0 = time.time(); na = nb = 0
    while time.time() - t0 < secs:
        if a:
            for _ in range(8): L.poseidon_batch_dev(t, n, d_in.data_ptr(), d_out.data_ptr(), None, sa.cuda_stream); na += 1
        if b:
            with torch.cuda.stream(sb):
                for _ in range(2): buf.zero_(); nb += 1
        torch.cuda.synchronize()
    dt = time.time() - t0
    print("%-10s %.2f s: poseidon launches/s %.1f  fills/s %.1f" % (name, dt, na / dt, nb / dt))
    return t0, time.time()

th = threading.Thread(target=sampler); th.start()
time.sleep(1.0)
spans = [("idle", time.time() - 1.0, time.time())]
for nm, a, b in (("valu", 1, 0), ("fill", 0, 1), ("both", 1, 1), ("valu2", 1, 0)):
    s = phase(nm, a, b); spans.append((nm,) + s)
stop = True; th.join()
for nm, s0, s1 in spans:
    print("==", nm)
    for ts, o in samples:
        if s0 <= ts <= s1: print("   ", o[-220:])
