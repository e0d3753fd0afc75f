"""GPU probe: which HIP streams of this process block each other through EVENT WAITS?

HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default).  Kernels of two streams that share a
queue still overlap (no barrier bit between packets of different streams); a `hipStreamWaitEvent` however is a barrier packet at
the head of its stream's hardware queue: until the event fires, NOTHING behind it in that queue runs -- including the other
stream's kernels.  That is what makes two decoder streams of the long-form loop 2 x slower than one in some processes
(profiles/LAB_NOTES.md round 6).  Probe of one ordered pair (a, s): a long spin kernel on a third stream, `a` waits for it, a tiny
kernel goes to `s`; s is blocked by a's waits iff the tiny kernel finishes only after the spin.
    python tools/probe_queues.py [streams=10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
if os.environ.get("PROBE_MAIN") == "pool":  # the caller's stream is a pool stream instead of the null stream
    torch.cuda.set_stream(ops.aux_stream(dev, 0, index=30))
main = torch.cuda.current_stream()
names = ["main"] + ["aux%d" % i for i in range(n)]
streams = [main] + [ops.aux_stream(dev, 0, index=i) for i in range(n)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
print("rows: the stream that waits (W) / runs a long kernel (K); columns: the stream whose tiny kernel is held up; capital = both trials")
print("%6s " % "" + " ".join("%5s" % s for s in names))
for i, a in enumerate(streams):
    row = []
    for j, s in enumerate(streams):
        if i == j:
            row.append("  -  ")
            continue
        w = [ops.wait_blocks(a, s) for _ in range(2)]
        k = [ops.wait_blocks(a, s, mode="kernel") for _ in range(2)]
        row.append(" %s%s  " % ("W" if all(w) else ("w" if any(w) else "."), "K" if all(k) else ("k" if any(k) else ".")))
    print("%6s " % names[i] + " ".join(row), flush=True)
