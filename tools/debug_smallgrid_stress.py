#!/usr/bin/env python
"""Are the small-grid builds of st2_conv1d_xs bitwise reproducible when another queue keeps the chip busy?  Stream A runs one conv
(with statistics) into a ring of outputs while stream B streams big activation passes / convs; every output and every statistics
tensor is compared with the unloaded reference afterwards."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from styletts2_amd import _lib, ops, weights  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
big = torch.randn(16, 256, 48000, generator=g).to(dev)
bw = weights.pack_conv_f16s(torch.randn(256, 256, 7, generator=g) / 40).to(dev)
side = torch.cuda.Stream()
for B, C, L, ks, dil in ((1, 256, 5680, 7, 1), (1, 128, 9600, 11, 5), (1, 256, 960, 3, 1), (1, 128, 14400, 7, 3), (2, 256, 2400, 7, 1)):
    x = torch.randn(B, C, L, generator=g).to(dev)
    w = weights.pack_conv_f16s(torch.randn(C, C, ks, generator=g) / math.sqrt(C * ks)).to(dev)
    res = torch.randn(B, C, L, generator=g).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    xs = ops.activate(x)
    kw = dict(dil=dil, pad_left=(ks - 1) * dil // 2, bias=bias, res=res, want_stats=True)
    ref, st_ref = ops.conv1d_xs(xs, w, C, ks, **kw)
    torch.cuda.synchronize()
    d = _lib.ConvDesc()
    d.B, d.C_in, d.C_out, d.L_in, d.L_out, d.ks = B, C, C, L, L, ks
    cols = _lib.load().st2_conv1d_xs_part_cols(d)
    for load in ("idle", "act passes on a second stream", "convs on a second stream"):
        outs = []
        torch.cuda.synchronize()
        if load != "idle":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(30 if load.startswith("act") else 12):
                    if load.startswith("act"):
                        ops.activate(big)
                    else:
                        ops.conv1d_xs(ops.activate(big[:4]), bw, 256, 7, pad_left=3)
        for _ in range(60):
            outs.append(ops.conv1d_xs(xs, w, C, ks, **kw))
        torch.cuda.synchronize()
        bad_y = sum(not torch.equal(o[0], ref) for o in outs)
        bad_s = sum(not torch.equal(o[1], st_ref) for o in outs)
        print("B %d C %3d L %5d k %2d d %d (tile columns %3d) %-32s: y differs in %2d / 60, statistics in %2d / 60" % (
            B, C, L, ks, dil, cols, load, bad_y, bad_s), flush=True)
