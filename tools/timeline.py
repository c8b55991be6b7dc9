"""Summarise a rocprofv3 kernel trace (sqlite .db or *_kernel_trace.csv): per-kernel totals, busy time (union of
kernel intervals) and a coarse text timeline. Usage: python tools/timeline.py <trace.db|trace.csv> [t0_ms t1_ms]"""
import csv, sqlite3, sys


def load(path):
    rows = []
    if path.endswith(".csv"):
        for r in csv.DictReader(open(path)):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")))
        return rows
    db = sqlite3.connect(path)
    tabs = [t[0] for t in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [c[1] for c in db.execute("pragma table_info(%s)" % kd)]
    q = "queue_id" if "queue_id" in cols else "0"
    for name, s, e, qu in db.execute("select k.kernel_name, d.start, d.end, d.%s from %s d join %s k on d.kernel_id = k.id" % (q, kd, ks)):
        rows.append((name, s, e, qu))
    return rows


def main():
    rows = sorted(load(sys.argv[1]), key=lambda r: r[1])
    t0 = rows[0][1]
    lo = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0
    hi = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 1e30
    rows = [r for r in rows if lo <= r[1] - t0 <= hi]
    short = lambda n: n.split("(")[0].replace("hz::", "").replace("void ", "")[:28]
    for n, s, e, q in rows:
        if e - s > 200000:
            print("%10.3f %10.3f %8.3f ms  q=%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, short(n)))


if __name__ == "__main__":
    main()
