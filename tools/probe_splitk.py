#!/usr/bin/env python
"""Split-K depth of the skinny fused convs (token GEMMs of one utterance: N ~ 100 columns): ms per launch of st2_conv1d_f16s for
the denoiser / PL-BERT Linear shapes at (max slices, min chunks per slice) = (8, 4) [default], (16, 2), (16, 1), (32, 1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from styletts2_amd import _hooks, _lib, ops, weights  # noqa: E402

lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
shapes = [(1024, 1024), (512, 1024), (1024, 512), (2048, 1024), (1024, 2048), (2304, 768), (768, 768), (2048, 768), (768, 2048), (512, 768)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
settings = [(8, 4), (16, 2), (16, 1), (32, 1)]
print("N = %d columns; us per launch (conv + reduction), 4 replays of a 50-launch graph" % N)
print("%-12s" % "M x K" + "".join("%12s" % ("%d/%d" % s) for s in settings))
with _hooks.override(conv_path="fused"):
    for M, K in shapes:
        x = torch.randn(1, K, N, generator=g).to(dev)
        wt = weights.pack_conv_f16s(torch.randn(M, K, 1, generator=g) / K ** 0.5).to(dev)
        bias = torch.randn(M, generator=g).to(dev)
        row = "%-12s" % ("%dx%d" % (M, K))
        ref = None
        for mx, mc in settings:
            lib.st2_conv1d_f16s_set_splitk(mx, mc)
            for _ in range(3):
                y = ops.conv1d(x, wt, M, 1, bias=bias)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()  # 50 launches per replay: the GPU's time, not the host's (18-23 us per eager call)
            with torch.cuda.graph(graph):
                for _ in range(50):
                    y = ops.conv1d(x, wt, M, 1, bias=bias)
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            err = (y - ref).abs().max().item() / ref.abs().max().item()
            row += "%9.1f%s" % (e0.elapsed_time(e1) / 200 * 1e3, " ok" if err < 1e-5 else " !!")
        print(row, flush=True)
lib.st2_conv1d_f16s_set_splitk(0, 0)
