"""GPU probe: does the act -> conv -> act -> conv hand-off of a vocoder resblock get cheaper when it stays inside the
256 MB Infinity Cache?  One AdaINResBlock1 chain (3 x [act+conv(dil d), act+conv(dil 1)+residual], statistics from the
conv epilogues) at C = 128, L = 48001 over 32 utterances, issued in sub-batches of S utterances (all six layers for
one sub-batch before the next one starts).  A [S, 128, 48001] fp32 tensor is S x 24.6 MB, its xs planes the same; the
chain keeps ~4 such tensors live.  Each variant is captured in a hipGraph so that launch overhead does not mask the
device time."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import ops, weights

dev = "cuda"
BT, Cc, L = 32, 128, int(os.environ.get("PROBE_L", "48001"))
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(BT, Cc, L, device=dev, generator=g)
H = torch.randn(BT, 2 * Cc, device=dev, generator=g) * 0.3
alpha = torch.rand(Cc, device=dev, generator=g) + 0.5
bias = torch.randn(Cc, device=dev, generator=g) * 0.1


def chain(x, h, wts, ks, dils):
    st = ops.instnorm_stats(x)
    for i, d in enumerate(dils):
        kw = dict(pro=ops.PRO_ADAIN_SNAKE, gamma=h[:, :Cc], beta=h[:, Cc:], alpha=alpha, bias=bias)
        xt, st2 = ops.conv1d(x, wts[2 * i], Cc, ks, dil=d, pad_left=(ks - 1) * d // 2, stats=st, want_stats=True, **kw)
        x, st = ops.conv1d(xt, wts[2 * i + 1], Cc, ks, dil=1, pad_left=(ks - 1) // 2, stats=st2, res=x, want_stats=True,
                           **kw)
    return x


for ks in (3, 7, 11):
    wts = [weights.pack_conv_f16s(torch.randn(Cc, Cc, ks, device=dev, generator=g) / math.sqrt(Cc * ks)).to(dev)
           for _ in range(6)]
    flop = 6 * 2.0 * BT * Cc * Cc * ks * L
    ref = None
    for S in (32, 16, 8, 4, 2):
        def run():
            outs = []
            for b0 in range(0, BT, S):
                outs.append(chain(X[b0:b0 + S], H[b0:b0 + S], wts, ks, (1, 3, 5)))
            return outs
        run()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            run()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            outs = run()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        out = torch.cat(outs)
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        print("ks=%2d sub-batch %2d: %.3f ms per 32-utterance resblock (%.1f algorithmic TFLOP/s incl. act passes), "
              "bitwise==S32: %s" % (ks, S, ms, flop / ms / 1e9, same), flush=True)
        del graph, outs, out
