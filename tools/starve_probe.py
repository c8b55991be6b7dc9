"""When can a tiny kernel on another stream run while the witness kernels of a step are on the device? Non-blocking probes:
one tiny kernel every 2 ms on stream p, each followed by an event; after the step the completion time of every probe is printed
next to its launch time (scheduling probe for the two-contexts-in-flight design)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from circuits_amd import lib, builder as B
L = lib()
bb = B.synthetic_batch(2048, 32, 256, 64, n_accounts=2048, seed=0x48455A31)
Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = L.ctx("rollup-main", nTx=2048, nLevels=32, maxL1Tx=256, maxFeeTx=64, n_instances=Bp)
c.set_inputs(bb.get_input(), instance=0)
for b in range(1, Bp):
    c.copy_instance_inputs(0, b)
s = torch.cuda.Stream()
dummies = [torch.cuda.Stream() for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 0)]   # shifts the stream -> hardware queue mapping
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(1, device="cuda")
p = torch.cuda.Stream(priority=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
x = torch.zeros(1024, device="cuda")
with torch.cuda.stream(p):   # create the probe stream's hardware queue before the measurement
    x.add_(1)
    torch.cuda.Event(enable_timing=True).record(p)
c.enqueue(s.cuda_stream); c.check()
torch.cuda.synchronize()
start = torch.cuda.Event(enable_timing=True)
start.record(s)
c.enqueue(s.cuda_stream)
t0 = time.perf_counter()
evs = []
while time.perf_counter() - t0 < 0.080:
    a = (time.perf_counter() - t0) * 1e3
    with torch.cuda.stream(p):
        x.add_(1)
        e = torch.cuda.Event(enable_timing=True); e.record(p)
    evs.append((a, e))
    time.sleep(0.002)
c.check()
torch.cuda.synchronize()
print("launched at (ms) -> completed at (ms after the step began):")
print(" ".join("%.0f>%.0f" % (a, start.elapsed_time(e)) for a, e in evs))
for name, ms, by, un in []:
    pass
