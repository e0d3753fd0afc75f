#!/usr/bin/env python
"""Overlapped synthesize_long with every st2_prosody_forward call issued TWICE on the same inputs (no sync in between): do the two
results agree bit for bit while the previous sentence's decoder shares the chip?  And the duration / front outputs?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_pipeline_gpu import KEYS, _model  # noqa: E402
from styletts2_amd import _lib, engine, models, pipeline  # noqa: E402

man, model, sds = _model("libritts")
g = torch.Generator().manual_seed(11)
lens, steps = [9, 6, 12, 7], 3
sentences = [torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, 178, (n - 1,), generator=g)]) for n in lens]
noises = [torch.randn(1, 1, 256, generator=g) for _ in lens]
step_noises = [torch.randn(steps - 1, 1, 1, 256, generator=g) for _ in lens]
durs = [torch.full((1, n), 2, dtype=torch.long) for n in lens]
sine = [torch.randn(1, 600 * 2 * n, 9, generator=g) for n in lens]
ref_s = torch.randn(1, 256, generator=g)
for k in KEYS:
    model[k].to("cuda")
sampler = models.make_sampler(model)
d = lambda xs: [x.to("cuda") for x in xs]
kw = dict(ref_s=ref_s.to("cuda"), t=0.7, diffusion_steps=steps, noises=d(noises), step_noises=d(step_noises), sine_noises=d(sine), durations=durs)
if len(sys.argv) > 1 and sys.argv[1] == "single":
    _lib.load().st2_lstm_coop_set_block(-1)
orig = engine.Engine.prosody_forward
pairs = []


def twice(self, *a, **k):
    r1 = orig(self, *a, **k)
    r2 = orig(self, *a, **k)
    pairs.append((r1, r2))
    return r1


engine.Engine.prosody_forward = twice
for trial in range(4):
    pairs.clear()
    pipeline.synthesize_long(model, sampler, d(sentences), overlap=True, **kw)
    torch.cuda.synchronize()
    print("trial %d:" % trial, " | ".join("s%d " % i + ",".join("%s:%s" % (n, "=" if torch.equal(a, b) else "%.0e" % (a - b).abs().max().item())
                                                               for n, a, b in zip(("asr", "F0", "N"), r1, r2)) for i, (r1, r2) in enumerate(pairs)), flush=True)
