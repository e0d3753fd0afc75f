#!/usr/bin/env python
"""When is ST2_STATUS_LSTM_RECOVERED raised?  Sentence-by-sentence synthesis (B = 1) and a B = 32 batch, eager front vs the
graph-replayed front: the sticky status word after every call, and whether the two fronts agree bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from benchdata import manifest, synth  # noqa: E402
from styletts2_amd import models, ops, pipeline  # noqa: E402

dev = torch.device("cuda", 0)
man = manifest("ljspeech")
model = bench.build(man)
for i, k in enumerate(bench.KEYS):
    synth.init_synthetic_(model[k], 10 + i)
    model[k].eval().to(dev)
sampler = models.make_sampler(model)
front = pipeline.GraphedFront(model, sampler)
for B, N in ((1, 100), (1, 64), (32, 100)):
    tokens, lengths, noise, durations, _ = bench.synthetic_inputs(B, 5)
    tokens, noise, dur = tokens[:, :N].contiguous().to(dev), noise.to(dev), durations[:, :N].contiguous().to(dev)
    lengths = lengths.clamp(max=N)
    step_noise = torch.randn(4, B, 1, 256, device=dev)
    outs = {}
    for mode, fr in (("eager", None), ("graph", front), ("graph", front), ("eager", None), ("graph", front)):
        ops.status(clear=True)
        p = pipeline.prepare(model, sampler, tokens, lengths, noise, diffusion_steps=5, durations=dur, total_frames=4 * N,
                             step_noise=step_noise, front=fr)
        torch.cuda.synchronize()
        st = ops.status(clear=True)
        key = (mode,)
        same = ""
        if "ref" in outs:
            same = " F0 equal to first run: %s, asr equal: %s" % (torch.equal(outs["ref"]["F0"], p["F0"]), torch.equal(outs["ref"]["asr"], p["asr"]))
        else:
            outs["ref"] = {k: p[k].clone() for k in ("F0", "asr")}
        print("B %2d N %3d %-5s front: status 0x%x%s" % (B, N, mode, st, same), flush=True)
