"""GPU probe: the ragged real-text batch (bench.py `ljspeech_ragged`) with its decoder calls on one chosen pair of auxiliary
streams -- wall time per step, and under `rocprofv3 --kernel-trace` the per-hardware-queue timeline of the steps.
    python tools/probe_ragged.py FIRST [steps=3] [eager]   (FIRST = index of the first auxiliary stream; 0 = the caller's stream only)
    PROBE_NSTREAMS=n: n consecutive auxiliary streams instead of 2"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import bench
from _util import manifest
from benchdata import synth
from styletts2_amd import models, ops, pipeline

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
man = manifest("ljspeech")
model = models.build_model(models.recursive_munch(man["config"]), None, None, models.load_plbert(man["plbert"]))
for i, k in enumerate(bench.KEYS):
    synth.init_synthetic_(model[k], 10 + i)
    model[k].eval().to(dev)
sampler = models.make_sampler(model, graph=False)
front = None if "eager" in sys.argv else pipeline.GraphedFront(model, sampler)
tokens, lengths, noise, dur, lens = bench.ragged_inputs(dev)
# the streams the bench process would have made before this leg (same creation order: front stream first, then the windows)
_ = ops.aux_stream(dev, -1), ops.aux_stream(dev, 0)
pool = [ops.aux_stream(dev, 0, index=i) for i in range(1, 9)]
NS = int(os.environ.get("PROBE_NSTREAMS", "2"))
streams = None if first == 0 else [ops.aux_stream(dev, 0, index=first + i) for i in range(NS)]


def step():
    return pipeline.inference(model, sampler, tokens, lengths, noise, diffusion_steps=5, embedding_scale=1.0, durations=dur,
                              front=front, decode_streams=streams)


for _ in range(2):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    t = time.perf_counter()
    step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
ops.check_status()
print("ragged batch, decoder streams %s: %s ms per step" % ("caller's" if streams is None else "aux%d..aux%d" % (first, first + NS - 1),
                                                            ", ".join("%.1f" % t for t in ts)))
