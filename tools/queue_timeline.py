"""Per-hardware-queue summary of a rocprofv3 kernel trace (csv): kernels, busy time and the largest idle gaps of every Queue_Id in
the last `frac` of the trace.  Usage: python tools/queue_timeline.py <kernel_trace.csv> [frac=0.3]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t_lo = rows[-1][1] - frac * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = (rows[-1][1] - rows[0][0]) / 1e6
print("window %.1f ms, %d kernels" % (span, len(rows)))
byq = defaultdict(list)
for r in rows:
    byq[(r[3], r[4])].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(e - s for s, e, *_ in rs) / 1e6
    gaps = sorted(((b[0] - a[1]) / 1e3, (a[1] - rows[0][0]) / 1e6, a[2][:40], b[2][:40]) for a, b in zip(rs, rs[1:]))
    big = [g for g in gaps if g[0] > 200]
    print("queue %s stream %s: %5d kernels, busy %.1f ms, first at %.1f ms, last at %.1f ms, %d gaps > 200 us (sum %.1f ms)" % (
        q[0], q[1], len(rs), busy, (rs[0][0] - rows[0][0]) / 1e6, (rs[-1][1] - rows[0][0]) / 1e6, len(big), sum(g[0] for g in big) / 1e3))
    for g in sorted(big, reverse=True)[:4]:
        print("      %.0f us idle at t = %.1f ms after %s before %s" % g)
