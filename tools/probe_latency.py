"""GPU probe: the B = 1 / 10 s latency point (bench.py `latency_b1_10s`) by itself -- wall time per call, and under
`rocprofv3 --kernel-trace` the kernel timeline of the calls (tools/trace_gaps.py: busy / idle time, per-kernel totals).
    python tools/probe_latency.py [calls=10]        PROBE_EAGER=1: eager front instead of the graph replay"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from _util import manifest
from benchdata import synth
from styletts2_amd import models, ops, pipeline

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
man = manifest("ljspeech")
model = models.build_model(models.recursive_munch(man["config"]), None, None, models.load_plbert(man["plbert"]))
KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]
for i, k in enumerate(KEYS):
    synth.init_synthetic_(model[k], 10 + i)
    model[k].eval().to(dev)
sampler = models.make_sampler(model, graph=False)
front = None if os.environ.get("PROBE_EAGER") else pipeline.GraphedFront(model, sampler)
g = torch.Generator().manual_seed(77)
N = 100
tokens = torch.randint(1, 178, (1, N), generator=g).to(dev)
lengths = torch.full((1,), N, dtype=torch.long)
noise = torch.randn(1, 1, 256, generator=g).to(dev)
dur = torch.full((1, N), 4, dtype=torch.long).to(dev)


def step():
    return pipeline.inference(model, sampler, tokens, lengths, noise, diffusion_steps=5, embedding_scale=1.0, durations=dur,
                              total_frames=4 * N, front=front)


for _ in range(4):
    step()
torch.cuda.synchronize()
ts, host = [], []
for _ in range(calls):
    torch.cuda.synchronize()
    t = time.perf_counter()
    step()
    host.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
ops.check_status()
print("B = 1, 10 s: latency min %.3f / mean %.3f ms; host returns after min %.3f / mean %.3f ms (%s front)"
      % (min(ts), sum(ts) / len(ts), min(host), sum(host) / len(host), "eager" if front is None else "graph-replayed"))
