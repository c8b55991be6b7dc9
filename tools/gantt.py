"""Text Gantt of a rocprofv3 kernel trace: one row per kernel name, one column per `dt` ms; the cell shows how many launches of that
kernel were running (1-9, + for more). Usage: python tools/gantt.py <kernel_trace.csv> t0_ms t1_ms [dt_ms]
Times are relative to the first kernel of the trace."""
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r[1])
    t0 = rows[0][1]
    lo, hi = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
    dt = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else 1e6
    nb = int((hi - lo) / dt)
    short = lambda n: n.split("(")[0].replace("hz::", "").replace("void ", "")[:22]
    names = []
    grid = {}
    for n, s, e in rows:
        s -= t0
        e -= t0
        if e < lo or s > hi:
            continue
        k = short(n)
        if k not in grid:
            grid[k] = [0] * nb
            names.append(k)
        b0, b1 = max(0, int((s - lo) / dt)), min(nb - 1, int((e - lo) / dt))
        for b in range(b0, b1 + 1):
            grid[k][b] += 1
    print("%-22s %s" % ("ms", "".join(str((int(lo / 1e6 + i * dt / 1e6) // 10) % 10) if i % 10 == 0 else " " for i in range(nb))))
    for k in names:
        print("%-22s %s" % (k, "".join(" " if c == 0 else (str(c) if c < 10 else "+") for c in grid[k])))


if __name__ == "__main__":
    main()
