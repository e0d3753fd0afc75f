#!/usr/bin/env python
"""LSTM outputs that differ under a small-grid conv load: WHERE do they differ (isolated elements = somebody else's stray store;
from some time step on in one direction = the recurrence itself took a wrong input)?  Also: guard bands around the conv's output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from styletts2_amd import _hooks, ops, weights  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
lx = ops.activate(torch.randn(1, 256, 5680, generator=g).to(dev))
lw = weights.pack_conv_f16s(torch.randn(256, 256, 7, generator=g) / 40).to(dev)
T = 24
G = torch.randn(1, 2048, T, generator=g).to(dev)
whh = (torch.randn(2, 256, 1024, generator=g) / 16).to(dev).contiguous()
side = torch.cuda.Stream()
# guard bands around the conv output
buf = torch.full((256 * 5680 + 2 * 65536,), 1234.5, device=dev)
yv = buf[65536:65536 + 256 * 5680].view(1, 256, 5680)
ops.conv1d_xs(lx, lw, 256, 7, pad_left=3, out=yv)
torch.cuda.synchronize()
print("guards around the small-grid conv output intact:", bool((buf[:65536] == 1234.5).all()) and bool((buf[-65536:] == 1234.5).all()))
for mode in ("single", "coop"):
    with _hooks.override(lstm=mode):
        ref = ops.lstm_bidir(G, whh).clone()
        torch.cuda.synchronize()
        for trial in range(3):
            outs = []
            side.wait_stream(torch.cuda.current_stream())
            for _ in range(150):
                ops.conv1d_xs(lx, lw, 256, 7, pad_left=3, out=yv)
            with torch.cuda.stream(side):
                for _ in range(40):
                    outs.append(ops.lstm_bidir(G, whh))
            torch.cuda.synchronize()
            for i, o in enumerate(outs):
                if not torch.equal(o, ref):
                    dmask = (o != ref)[0]                      # [2H, T]
                    fwd, rev = dmask[:256], dmask[256:]
                    tf = fwd.any(0).nonzero().flatten().tolist()
                    tr = rev.any(0).nonzero().flatten().tolist()
                    print("%s trial %d call %2d: %5d elements differ, max %.2e; forward rows differ at t = %s (units %d), reverse at t = %s (units %d)" % (
                        mode, trial, i, int(dmask.sum()), (o - ref).abs().max().item(), tf[:6] + (["..."] if len(tf) > 6 else []),
                        int(fwd.any(1).sum()), tr[:6] + (["..."] if len(tr) > 6 else []), int(rev.any(1).sum())), flush=True)
                    break
print("guards still intact:", bool((buf[:65536] == 1234.5).all()) and bool((buf[-65536:] == 1234.5).all()))
