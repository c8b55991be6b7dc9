"""Runs mfmabench for each unit while sampling the package power with rocm-smi; prints rate, mean power, energy per byte product."""
import os, re, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
def sample(stop, out):
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "-P", "--csv"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5).stdout
            m = re.findall(r"card0,([0-9.]+)", o)
            if m: out.append((time.time(), float(m[0])))
        except Exception:
            pass
        time.sleep(0.25)
def idle_power():
    out, stop = [], [False]
    th = threading.Thread(target=sample, args=(stop, out), daemon=True); th.start(); time.sleep(2.0); stop[0] = True; th.join()
    return sum(p for _, p in out) / max(1, len(out))
idle = idle_power()
print("idle package power %.0f W" % idle)
for unit in ("valu", "mfma32", "mfma16"):
    out, stop = [], [False]
    th = threading.Thread(target=sample, args=(stop, out), daemon=True); th.start()   # daemon: a failure below must not leave the sampler running
    t0 = time.time()
    r = subprocess.run([os.path.join(HERE, "mfmabench"), unit, "4"], stdout=subprocess.PIPE, text=True).stdout.strip()
    t1 = time.time(); stop[0] = True; th.join()
    ps = [p for t, p in out if t0 + 1.0 < t < t1 - 0.3]
    w = sum(ps) / max(1, len(ps))
    rate = float(re.search(r"([0-9.e+]+) byte-MAC/s", r).group(1))
    print(r)
    print("   package power %.0f W (%d samples, max %.0f) -> %.2f pJ per byte-MAC at the package, %.2f pJ above idle" % (w, len(ps), max(ps or [0]), w / rate * 1e12, (w - idle) / rate * 1e12))
