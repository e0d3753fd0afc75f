"""Busy / idle / overlapped GPU time from a rocprofv3 kernel trace (csv): is a schedule slow because its kernels got slower
(interference between queues) or because the queues wait for each other (gaps)?  Looks at the last 40 % of the trace
(steady-state steps).  Usage: python tools/trace_gaps.py <..._kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    t_lo = rows[0][0] + 0.6 * (rows[-1][1] - rows[0][0])
    rows = [r for r in rows if r[0] >= t_lo]
    span = rows[-1][1] - rows[0][0]
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy = over = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    per = defaultdict(lambda: [0, 0])
    for s, e, n, _ in rows:
        k = n.split("(")[0][-60:]
        per[k][0] += 1
        per[k][1] += e - s
    queues = sorted({r[3] for r in rows})
    print("trace window %.1f ms, %d kernels on queues %s" % (span / 1e6, len(rows), queues))
    print("  >= 1 kernel running %.1f ms (%.1f %%), idle %.1f ms, >= 2 running %.1f ms; sum of kernel durations %.1f ms" % (
        busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, over / 1e6, sum(v[1] for v in per.values()) / 1e6))
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %8.3f ms  %5d x %8.1f us  %s" % (t / 1e6, n, t / n / 1e3, k))


if __name__ == "__main__":
    main(sys.argv[1])
