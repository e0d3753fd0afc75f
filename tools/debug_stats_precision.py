#!/usr/bin/env python
"""Are the InstanceNorm statistics that ride in the conv epilogues (per-tile fp32 partial sums + fp64 finalize) as good as a
direct fp64 reduction of the stored tensor, on a small-magnitude checkpoint?  Python per-kernel plan of the decoder with every
want_stats conv / convt_interleave call checked against st2_instnorm_stats of its own output: the discrepancy is reported in
units that matter to the consuming AdaIN, |d mean| * rstd and |d rstd| / rstd."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _util import decoder_kwargs, manifest  # noqa: E402
from benchdata import synth  # noqa: E402
from styletts2_amd import _hooks, ops  # noqa: E402
from styletts2_amd.decoder import Decoder  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "libritts"
f = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
dc = manifest(tag)["config"]["decoder"]
dec = Decoder(**decoder_kwargs(dc)).eval()
synth.init_trained_like_(dec, 1)
synth.scale_params_(dec, {"decode.3.conv2.": f, "decode.3.conv1x1.": f, "generator.ups.": 0.1, "generator.noise_convs.": f, ".convs2.": f})
asr, F0, N, s, noise = synth.decoder_inputs(2, 24, 3)
dec = dec.cuda()
rows = []


def check(kind, out, st):
    ex = ops.instnorm_stats(out)
    dm = ((st[..., 0] - ex[..., 0]).abs() * ex[..., 1]).max().item()
    dr = ((st[..., 1] - ex[..., 1]).abs() / ex[..., 1]).max().item()
    x = out.double()
    ratio = (x.mean(-1).abs() / x.std(-1).clamp(min=1e-30)).max().item()
    rows.append((kind, tuple(out.shape), dm, dr, ratio, float(ex[..., 1].max())))


orig_conv1d, orig_xs = ops.conv1d, ops.conv1d_xs


def conv1d(*a, **k):
    r = orig_conv1d(*a, **k)
    if k.get("want_stats"):
        check("conv1d (fused or xs)", r[0], r[1])
    return r


ops.conv1d = conv1d
import styletts2_amd.decoder as D  # noqa: E402
names = [n for n in dir(ops) if "interleave" in n]
print("interleave entry points:", names)
for n in names:
    fn = getattr(ops, n)

    def wrap(*a, _fn=fn, _n=n, **k):
        r = _fn(*a, **k)
        if isinstance(r, tuple) and len(r) == 2 and torch.is_tensor(r[1]) and r[1].shape[-1] == 2:
            check(_n, r[0], r[1])
        return r
    setattr(ops, n, wrap)
with _hooks.override(plan="python"):
    dec._pk = None
    dec(asr.cuda() * f, F0.cuda(), N.cuda(), s.cuda(), noise=noise.cuda())
torch.cuda.synchronize()
rows.sort(key=lambda r: -max(r[2], r[3]))
print("%-24s %-18s %12s %12s %10s %10s" % ("producer", "tensor", "|dmean|*rstd", "|drstd|/rstd", "|mean|/std", "rstd max"))
for r in rows[:14]:
    print("%-24s %-18s %12.2e %12.2e %10.1f %10.1f" % (r[0], r[1], r[2], r[3], r[4], r[5]))
print("%d statistics checked" % len(rows))
