"""Are two HIP streams independent hardware queues? A long spin kernel on one stream, a tiny kernel on another."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "16")
import torch
x = torch.zeros(1024, device="cuda")
streams = [torch.cuda.Stream() for _ in range(12)]
torch.cuda._sleep(1000); torch.cuda.synchronize()
for k in (1, 2, 3, 5, 8, 11):
    with torch.cuda.stream(streams[0]):
        torch.cuda._sleep(int(60e6))   # ~25-30 ms spin, one block
    t = time.perf_counter()
    with torch.cuda.stream(streams[k]):
        x.add_(1)
    streams[k].synchronize()
    dt = (time.perf_counter() - t) * 1e3
    torch.cuda.synchronize()
    print("GPU_MAX_HW_QUEUES=%s stream %d behind a spinning stream 0: %.2f ms" % (os.environ["GPU_MAX_HW_QUEUES"], k, dt))
