"""From a rocprofv3 kernel trace (csv) of a whole bench.py run: the stretches in which kernels run on TWO auxiliary streams besides
the busiest one (the decoder-stream windows of the long-form and ragged legs), with each stream's hardware queue, kernel count, busy
time, the time both decode streams run a kernel at once, and the longest stalls.  Usage: python tools/stream_windows.py <kernel_trace.csv>"""
import csv
import sys
from collections import Counter, defaultdict

rows = []
with open(sys.argv[1]) as f:
    rd = csv.DictReader(f)
    has_stream = "Stream_Id" in rd.fieldnames
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Stream_Id"] if has_stream else r["Queue_Id"],
                     r["Kernel_Name"]))
rows.sort()
t0 = rows[0][0]
print("%d kernels, %.1f s, Stream_Id column: %s" % (len(rows), (rows[-1][1] - t0) / 1e9, has_stream))
print("streams (kernels):", Counter(r[3] for r in rows).most_common(12))
print("queues  (kernels):", Counter(r[2] for r in rows).most_common(12))
print("stream -> queues:", {s: sorted({r[2] for r in rows if r[3] == s}) for s in {r[3] for r in rows}})
# 20 ms bins: which streams are active
BIN = 20e6
bins = defaultdict(Counter)
for s, e, q, st, n in rows:
    bins[int((s - t0) // BIN)][st] += 1
main = Counter(r[3] for r in rows).most_common(1)[0][0]
segs = []
for b in sorted(bins):
    act = tuple(sorted(st for st, c in bins[b].items() if c >= 20 and st != main))
    if segs and segs[-1][0] == act and segs[-1][2] == b - 1:
        segs[-1][2] = b
    else:
        segs.append([act, b, b])
for act, b0, b1 in segs:
    if len(act) < 2 or b1 - b0 < 5:
        continue
    lo, hi = t0 + b0 * BIN, t0 + (b1 + 1) * BIN
    sel = [r for r in rows if lo <= r[0] < hi]
    print("t = %.2f .. %.2f s (%.0f ms): streams %s" % ((lo - t0) / 1e9, (hi - t0) / 1e9, (hi - lo) / 1e6, act))
    ev = []
    for st in act + (main,):
        rs = [r for r in sel if r[3] == st]
        busy = sum(r[1] - r[0] for r in rs) / 1e6
        gaps = sorted(((b[0] - a[1]) / 1e3 for a, b in zip(rs, rs[1:])), reverse=True)
        print("    stream %s on queue(s) %s: %6d kernels, busy %7.1f ms, gaps > 0.5 ms: %d (sum %.1f ms, largest %s)" % (
            st, sorted({r[2] for r in rs}), len(rs), busy, sum(1 for g in gaps if g > 500), sum(g for g in gaps if g > 500) / 1e3,
            ["%.1f" % (g / 1e3) for g in gaps[:3]]))
        if st != main:
            ev += [(r[0], 1) for r in rs] + [(r[1], -1) for r in rs]
    ev.sort()
    depth, last, both, any_ = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1:
            any_ += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    print("    decode streams: >= 1 kernel running %.1f ms, both at once %.1f ms" % (any_ / 1e6, both / 1e6))
