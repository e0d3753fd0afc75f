#!/usr/bin/env python
"""synthesize_long overlapped vs sequential: which sentences differ, by how much, and what does the status word say?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_pipeline_gpu import KEYS, _model  # noqa: E402
from styletts2_amd import _hooks, _lib, models, ops, pipeline  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "libritts"
man, model, sds = _model(tag)
g = torch.Generator().manual_seed(11)
lens, steps = [9, 6, 12, 7], 3
sentences = [torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, 178, (n - 1,), generator=g)]) for n in lens]
noises = [torch.randn(1, 1, 256, generator=g) for _ in lens]
step_noises = [torch.randn(steps - 1, 1, 1, 256, generator=g) for _ in lens]
durs = [torch.full((1, n), 2, dtype=torch.long) for n in lens]
sine = [torch.randn(1, 600 * 2 * n, 9, generator=g) for n in lens]
ref_s = torch.randn(1, 256, generator=g) if man["config"]["multispeaker"] else None
for k in KEYS:
    model[k].to("cuda")
sampler = models.make_sampler(model)
d = lambda xs: [x.to("cuda") for x in xs]
kw = dict(ref_s=None if ref_s is None else ref_s.to("cuda"), t=0.7, diffusion_steps=steps, noises=d(noises), step_noises=d(step_noises),
          sine_noises=d(sine), durations=durs)
runs = {}
for name, ovl in (("seq1", False), ("seq2", False), ("ovl1", True), ("ovl2", True), ("seq3", False)):
    ops.status(clear=True)
    waves, style = pipeline.synthesize_long(model, sampler, d(sentences), overlap=ovl, **kw)
    torch.cuda.synchronize()
    runs[name] = [w.clone() for w in waves]
    print("%s: status 0x%x" % (name, ops.status(clear=True)), flush=True)
ref = runs["seq1"]
for name, ws in runs.items():
    print(name, ["equal" if torch.equal(a, b) else "%.2e" % (a - b).abs().max().item() for a, b in zip(ws, ref)])
lib = _lib.load()
lib.st2_lstm_coop_set_block(-1)  # no cooperative launches at all
runs2 = {}
for name, ovl in (("seq/single-CU lstm", False), ("ovl/single-CU lstm", True)):
    waves, style = pipeline.synthesize_long(model, sampler, d(sentences), overlap=ovl, **kw)
    torch.cuda.synchronize()
    runs2[name] = [w.clone() for w in waves]
lib.st2_lstm_coop_set_block(0)
print("single-CU lstm: ovl vs seq", ["equal" if torch.equal(a, b) else "%.2e" % (a - b).abs().max().item()
                                    for a, b in zip(runs2["ovl/single-CU lstm"], runs2["seq/single-CU lstm"])])

# ---- where does it differ: the decoder's inputs (front affected by the concurrent decoder) or its output for equal inputs? -------
lib.st2_lstm_coop_set_block(-1)
rec = {}
orig = model.decoder.forward


def spy(asr, F0, N, s, noise=None, **k):
    out = orig(asr, F0, N, s, noise=noise, **k)
    rec.setdefault(mode, []).append(dict(asr=asr.clone(), F0=F0.clone(), N=N.clone(), s=s.clone(), out=out.clone()))
    return out


model.decoder.forward = spy
for mode, ovl in (("seq", False), ("ovl", True)):
    pipeline.synthesize_long(model, sampler, d(sentences), overlap=ovl, **kw)
    torch.cuda.synchronize()
model.decoder.forward = orig
for k in range(len(lens)):
    a, b = rec["seq"][k], rec["ovl"][k]
    print("sentence %d:" % k, {n: ("equal" if torch.equal(a[n], b[n]) else "%.2e" % (a[n] - b[n]).abs().max().item()) for n in a})
lib.st2_lstm_coop_set_block(0)
