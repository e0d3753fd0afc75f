"""Summary of an xs_bench_64 per-workgroup timeline (one line per workgroup: id, HW_ID words, s_memtime at start / end of k loop /
end, s_memrealtime at start / end): launch span, start skew, k-loop and epilogue durations, workgroups per CU.
Usage: python tools/tl_summary.py xs_timeline.txt"""
import sys
from collections import Counter

rows = []
for line in open(sys.argv[1]):
    p = line.split()
    rows.append((int(p[0]), int(p[1], 16), int(p[2]), int(p[3]), int(p[4]), int(p[5]), int(p[6])))
r0 = min(r[5] for r in rows)
r1 = max(r[6] for r in rows)
print("%d workgroups; launch span %.1f us (s_memrealtime, 100 MHz)" % (len(rows), (r1 - r0) / 100.0))
starts = sorted((r[5] - r0) / 100.0 for r in rows)
print("start times: median %.1f us, 90 %% %.1f us, last %.1f us" % (starts[len(starts) // 2], starts[int(len(starts) * 0.9)], starts[-1]))
dur = sorted((r[6] - r[5]) / 100.0 for r in rows)
print("workgroup duration: min %.1f, median %.1f, max %.1f us" % (dur[0], dur[len(dur) // 2], dur[-1]))
kl = sorted((r[3] - r[2]) for r in rows)
ep = sorted((r[4] - r[3]) for r in rows)
print("k loop (s_memtime ticks): min %d, median %d, max %d; epilogue: min %d, median %d, max %d" % (kl[0], kl[len(kl) // 2], kl[-1], ep[0], ep[len(ep) // 2], ep[-1]))
hw = Counter()
for r in rows:
    w0 = r[1] & 0xffffffff
    cu = (w0 >> 8) & 0xf
    sh = (w0 >> 12) & 0x1
    se = (w0 >> 13) & 0x7
    xcc = (r[1] >> 32) & 0xf
    hw[(xcc, se, sh, cu)] += 1
print("CUs used: %d; workgroups per CU: %s" % (len(hw), sorted(Counter(hw.values()).items())))
