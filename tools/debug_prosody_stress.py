#!/usr/bin/env python
"""Which kernel of the prosody path is not reproducible while small-grid xs convs run on another stream?  Each candidate op runs 40 x
on a side stream under the load and every result is compared with its idle reference."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from styletts2_amd import _hooks, _lib, ops, weights  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
lib = _lib.load()
# the load: small-grid convs (B = 1, C = 256, L = 5 680: 32-column tiles) on the main stream
lx = ops.activate(torch.randn(1, 256, 5680, generator=g).to(dev))
lw = weights.pack_conv_f16s(torch.randn(256, 256, 7, generator=g) / 40).to(dev)
bigx = ops.activate(torch.randn(8, 128, 48000, generator=g).to(dev))
bigw = weights.pack_conv_f16s(torch.randn(128, 128, 7, generator=g) / 30).to(dev)


def load_small():
    for _ in range(150):
        ops.conv1d_xs(lx, lw, 256, 7, pad_left=3)


def load_big():
    for _ in range(6):
        ops.conv1d_xs(bigx, bigw, 128, 7, pad_left=3)


side = torch.cuda.Stream()
T = 24
C = 512
x = torch.randn(1, C, T, generator=g).to(dev)
x2 = torch.randn(1, C, 2 * T, generator=g).to(dev)
h = (torch.randn(1, 2 * C, generator=g) * 0.3).to(dev)
w3 = weights.pack_conv_f16s(torch.randn(C, C, 3, generator=g) / math.sqrt(3 * C)).to(dev)
w1 = weights.pack_conv_f16s(torch.randn(2048, 640, 1, generator=g) / math.sqrt(640)).to(dev)
xl = torch.randn(1, 640, T, generator=g).to(dev)
bias = torch.randn(C, generator=g).to(dev)
st = ops.instnorm_stats(x2)
G = torch.randn(1, 2048, T, generator=g).to(dev)
whh = (torch.randn(2, 256, 1024, generator=g) / 16).to(dev).contiguous()
sv = torch.randn(1, 128, generator=g).to(dev)
fcw = torch.randn(128, 6000, generator=g).to(dev)
fcb = torch.randn(6000, generator=g).to(dev)

cands = {
    "fused conv k3 + AdaIN + statistics": lambda: ops.conv1d(x2, w3, C, 3, pad_left=1, bias=bias, pro=ops.PRO_ADAIN_LEAKY, slope=0.2, stats=st,
                                                              gamma=h[:, :C], beta=h[:, C:], want_stats=True),
    "fused conv k3 + AdaIN + residual": lambda: ops.conv1d(x2, w3, C, 3, pad_left=1, bias=bias, pro=ops.PRO_ADAIN_LEAKY, slope=0.2, stats=st,
                                                            gamma=h[:, :C], beta=h[:, C:], res=x2, div=math.sqrt(2.0)),
    "fused conv k1 split-K (LSTM input projection)": lambda: ops.conv1d(xl, w1, 2048, 1),
    "instnorm_stats": lambda: ops.instnorm_stats(x2),
    "lstm single-CU": None, "lstm cooperative": None,
    "style_fc": lambda: ops.style_fc(sv, fcw, fcb),
}


def lstm(mode):
    with _hooks.override(lstm=mode):
        return ops.lstm_bidir(G, whh)


cands["lstm single-CU"] = lambda: lstm("single")
cands["lstm cooperative"] = lambda: lstm("coop")
flat = lambda r: [t for t in (r if isinstance(r, tuple) else (r,))]
with _hooks.override(conv_path="fused"):
    for name, fn in cands.items():
        ref = [t.clone() for t in flat(fn())]
        torch.cuda.synchronize()
        line = "%-48s" % name
        for lname, load in (("idle", None), ("small-grid convs", load_small), ("big convs", load_big)):
            outs = []
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            if load is not None:
                load()
            with torch.cuda.stream(side):
                for _ in range(40):
                    outs.append(flat(fn()))
            torch.cuda.synchronize()
            bad = sum(any(not torch.equal(a, b) for a, b in zip(o, ref)) for o in outs)
            line += "  %s: %2d / 40 differ" % (lname, bad)
        print(line, flush=True)
