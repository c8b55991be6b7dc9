"""Which kind of big kernel starves a tiny kernel on another stream: a persistent grid-stride one (hz_poseidon_batch_dev: 2048 blocks)
or one with thousands of pending workgroups (k_withdraw / k_withdraw_sha)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from circuits_amd import lib
L = lib()
x = torch.zeros(1024, device="cuda")
s, p = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(p):
    x.add_(1)
torch.cuda.synchronize()

def probe(label, launch, dur_ms):
    launch(); torch.cuda.synchronize()
    launch()
    t0 = time.perf_counter(); lat = []
    while time.perf_counter() - t0 < dur_ms * 1e-3:
        a = time.perf_counter()
        with torch.cuda.stream(p):
            x.add_(1)
        p.synchronize()
        lat.append("%.0f:%.2f" % ((a - t0) * 1e3, (time.perf_counter() - a) * 1e3))
        time.sleep(0.003)
    torch.cuda.synchronize()
    print(label, " ".join(lat[:12]))

n = 1 << 22
t = 3
d_in = torch.randint(0, 2**31 - 1, (n * (t - 1), 8), dtype=torch.int32).cuda(); d_in[:, 7] &= 0x0FFFFFFF
d_out = torch.empty((n, 8), dtype=torch.int32, device="cuda")
probe("poseidon digest 2^22 (persistent grid):", lambda: L.poseidon_batch_dev(t, n, d_in.data_ptr(), d_out.data_ptr(), None, s.cuda_stream), 20)
N = 1 << 15
c = L.ctx("withdraw", nLevels=32, n_instances=N)
for name, ln in c.input_names():
    c.set_input(name, [[0] * (ln) if ln > 1 else 0 for _ in range(N)] if ln > 1 else [0] * N, instance=-1)
def w():
    c.enqueue(s.cuda_stream)
probe("withdraw 2^15 (512 + 1024 workgroups):", w, 40)
try:
    c.check()
except Exception:
    pass
