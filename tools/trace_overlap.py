#!/usr/bin/env python
"""Answers from a `rocprofv3 --kernel-trace` CSV (timestamps per dispatch) the questions a `--stats` summary cannot:
do the two HIP queues of the pipelined bench (front of step k+1 / decoder of step k) execute concurrently on this box,
and what do the cooperative BiLSTM launches and the big convs cost when they do.

    python tools/trace_overlap.py <kernel_trace.csv> [--skip-ms 0] > summary.json

Per queue: dispatches, busy time (union of its kernels' intervals).  Across queues: time during which kernels of >= 2
queues are in flight, as a share of the busiest queue's busy time.  Per kernel family: count, mean / min / max duration,
split by whether another queue had a kernel in flight during the dispatch.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"(conv1d_xs_kernel(?:_o3)?<\d+|conv1d_f16s_kernel<\d+|lstm_coop_kernel|act_split_kernel<\d+|"
                  r"attention_kernel|style_fc_kernel|convt_interleave\w*|Cijk|copyBuffer|at::native)", name)
    return m.group(1) if m else name.split("(")[0][-40:]


def union_len(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    path = sys.argv[1]
    skip_ms = float(sys.argv[sys.argv.index("--skip-ms") + 1]) if "--skip-ms" in sys.argv else 0.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
            rows.append((s, e, str(q), r["Kernel_Name"]))
    rows.sort()
    t0 = rows[0][0] + int(skip_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    by_q = defaultdict(list)
    for s, e, q, _ in rows:
        by_q[q].append((s, e))
    queues = {q: {"dispatches": len(iv), "busy_ms": union_len(iv) / 1e6} for q, iv in by_q.items()}
    # sweep: time with >= 2 queues active
    ev = []
    for s, e, q, _ in rows:
        ev.append((s, 1, q))
        ev.append((e, -1, q))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = defaultdict(int)
    last, multi, any_t = None, 0, 0
    for t, d, q in ev:
        if last is not None:
            n = sum(1 for v in active.values() if v > 0)
            if n >= 2:
                multi += t - last
            if n >= 1:
                any_t += t - last
        active[q] += d
        last = t
    # per-family durations split by cross-queue overlap (does any kernel of ANOTHER queue intersect the dispatch?)
    other = {q: sorted(iv for qq, ivs in by_q.items() if qq != q for iv in ivs) for q in by_q}
    import bisect
    starts = {q: [iv[0] for iv in other[q]] for q in by_q}
    # running max of end times for a correct intersect test on start-sorted intervals
    maxend = {}
    for q in by_q:
        m, acc = [], 0
        for s, e in other[q]:
            acc = max(acc, e)
            m.append(acc)
        maxend[q] = m
    fam = defaultdict(lambda: {"alone": [], "overlapped": []})
    for s, e, q, name in rows:
        i = bisect.bisect_left(starts[q], e)  # intervals starting before this one ends
        ov = i > 0 and maxend[q][i - 1] > s
        fam[family(name)]["overlapped" if ov else "alone"].append((e - s) / 1e3)
    out_f = {}
    for k, v in fam.items():
        tot = sum(v["alone"]) + sum(v["overlapped"])
        d = {"total_ms": tot / 1e3}
        for kk in ("alone", "overlapped"):
            x = v[kk]
            if x:
                d[kk] = {"n": len(x), "mean_us": sum(x) / len(x), "min_us": min(x), "max_us": max(x)}
        out_f[k] = d
    out_f = dict(sorted(out_f.items(), key=lambda kv: -kv[1]["total_ms"])[:25])
    span = (rows[-1][1] - rows[0][0]) / 1e6
    busiest = max(q["busy_ms"] for q in queues.values())
    print(json.dumps({"span_ms": span, "gpu_busy_ms": any_t / 1e6, "queues": queues,
                      "two_or_more_queues_active_ms": multi / 1e6,
                      "overlap_share_of_busiest_queue": multi / 1e6 / busiest if busiest else None,
                      "families": out_f}, indent=1))


if __name__ == "__main__":
    main()
