#!/usr/bin/env python
"""Overlapped vs sequential synthesize_long with the single-CU LSTM (the flaky case): keep every prepare() result alive (references
only: no clone, no sync) and compare the decoder inputs and the waveforms per sentence afterwards."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_pipeline_gpu import KEYS, _model  # noqa: E402
from styletts2_amd import _lib, models, ops, pipeline  # noqa: E402

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
lib = _lib.load()
lib.st2_lstm_coop_set_block(-1)
keep_refs = len(sys.argv) > 1 and sys.argv[1] == "keep"
orig_prepare = pipeline.prepare
for trial in range(4):
    res = {}
    for mode, ovl in (("seq", False), ("ovl", True)):
        kept = []

        def spy(*a, **k):
            p = orig_prepare(*a, **k)
            if keep_refs:
                kept.append(p)
            return p
        pipeline.prepare = spy
        waves, style = pipeline.synthesize_long(model, sampler, d(sentences), overlap=ovl, **kw)
        torch.cuda.synchronize()
        pipeline.prepare = orig_prepare
        res[mode] = (waves, kept)
    line = "trial %d (%s): waves " % (trial, "references kept" if keep_refs else "nothing kept")
    line += str(["equal" if torch.equal(a, b) else "%.1e" % (a - b).abs().max().item() for a, b in zip(res["seq"][0], res["ovl"][0])])
    if keep_refs:
        for k in range(len(lens)):
            a, b = res["seq"][1][k], res["ovl"][1][k]
            line += " | s%d " % k + ",".join("%s:%s" % (n, "=" if torch.equal(a[n], b[n]) else "%.0e" % (a[n] - b[n]).abs().max().item())
                                              for n in ("asr", "F0", "N", "ref"))
    print(line, flush=True)
