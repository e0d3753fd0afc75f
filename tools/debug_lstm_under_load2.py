#!/usr/bin/env python
"""Which property of the other queue's work makes the BiLSTM kernels irreproducible?  Loads: the same small conv in 32- / 64- / 128-column
tiles (want_stats + part_cols forces the width), many small activation passes, many small fused convs, a big conv.
The k = 7 / 11 narrow builds are compiled only with -DST2_XS_NARROW_ALL=1 (profiles/LAB_NOTES.md round 5): build the library with that flag
(styletts2_amd/_build.py FLAGS) to reproduce the finding; with the product library those loads are reported as "not built"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from styletts2_amd import _hooks, ops, weights  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
x_small = torch.randn(1, 256, 5680, generator=g).to(dev)
lx = ops.activate(x_small)
lw = weights.pack_conv_f16s(torch.randn(256, 256, 7, generator=g) / 40).to(dev)
lw11 = weights.pack_conv_f16s(torch.randn(256, 256, 11, generator=g) / 50).to(dev)
lw3 = weights.pack_conv_f16s(torch.randn(256, 256, 3, generator=g) / 30).to(dev)
bigx = ops.activate(torch.randn(8, 128, 48000, generator=g).to(dev))
bigw = weights.pack_conv_f16s(torch.randn(128, 128, 7, generator=g) / 30).to(dev)
xf = torch.randn(1, 64, 5680, generator=g).to(dev)
wf = weights.pack_conv_f16s(torch.randn(64, 64, 7, generator=g) / 20).to(dev)
T = 24
G = torch.randn(1, 2048, T, generator=g).to(dev)
whh = (torch.randn(2, 256, 1024, generator=g) / 16).to(dev).contiguous()
side = torch.cuda.Stream()
y = torch.empty(1, 256, 5680, device=dev)


def conv_cols(cols, n=150, w=lw, ks=7):
    return lambda: [ops.conv1d_xs(lx, w, 256, ks, pad_left=(ks - 1) // 2, out=y, want_stats=True, part_cols=cols) for _ in range(n)]


loads = {
    "idle": lambda: None,
    "k7 conv, 32-column tiles x150": conv_cols(32),
    "k7 conv, 64-column tiles x150": conv_cols(64),
    "k7 conv, 128-column tiles x100": conv_cols(128, 100),
    "k3 conv, 32-column tiles x200": conv_cols(32, 200, lw3, 3),
    "k11 conv, 32-column tiles x120": conv_cols(32, 120, lw11, 11),
    "k11 conv, 128-column tiles x80": conv_cols(128, 80, lw11, 11),
    "small activation passes x400": lambda: [ops.activate(x_small) for _ in range(400)],
    "small fused convs (C = 64) x150": lambda: [ops.conv1d(xf, wf, 64, 7, pad_left=3) for _ in range(150)],
    "big conv x6": lambda: [ops.conv1d_xs(bigx, bigw, 128, 7, pad_left=3) for _ in range(6)],
}
for mode in ("coop", "single"):
    with _hooks.override(lstm=mode, conv_path="fused"):
        ref = ops.lstm_bidir(G, whh).clone()
        torch.cuda.synchronize()
        for name, load in loads.items():
            bad = tot = 0
            for trial in range(3):
                outs = []
                side.wait_stream(torch.cuda.current_stream())
                try:
                    load()
                except Exception as e:  # the product library refuses part_cols = 32 / 64 at k = 7 / 11
                    print("%-7s lstm under %-34s: not built (%s)" % (mode, name, str(e)[:60]))
                    tot = -1
                    break
                with torch.cuda.stream(side):
                    for _ in range(30 if mode == "coop" else 10):
                        outs.append(ops.lstm_bidir(G, whh))
                torch.cuda.synchronize()
                bad += sum(not torch.equal(o, ref) for o in outs)
                tot += len(outs)
            if tot > 0:
                print("%-7s lstm under %-34s: %3d / %3d calls differ" % (mode, name, bad, tot), flush=True)
