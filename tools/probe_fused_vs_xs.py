"""GPU probe: for the narrow vocoder layers (C = 64 at L = 120 000, C = 32 at L = 240 000, B = 32), the fused kernel
(st2_conv1d_f16s: prologue inside the conv) against the pair (st2_act_split + st2_conv1d_xs) -- which side of the
ops.prefer_fused rule each kernel size belongs on today."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _hooks, ops, weights

dev = "cuda"
torch.manual_seed(0)


def timed(fn, n=4):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = 32
SHAPES = ((64, 120000), (32, 240000), (128, 48001))
if os.environ.get("PROBE_C"):  # e.g. PROBE_C=128 PROBE_KS=3: one side of the rule only
    SHAPES = tuple(s for s in SHAPES if s[0] == int(os.environ["PROBE_C"]))
KS = tuple(int(k) for k in os.environ.get("PROBE_KS", "3,7,11").split(","))
for (C, L) in SHAPES:
    pitch = (L + 31) // 32 * 32
    x = torch.randn(B, C, pitch, device=dev)[:, :, :L]
    out = torch.empty((B, C, pitch), device=dev)[:, :, :L]
    st = ops.instnorm_stats(x)
    h = torch.randn(B, 2 * C, device=dev) * 0.3
    alpha = torch.rand(C, device=dev) + 0.5
    bias = torch.randn(C, device=dev)
    for ks in KS:
        w = torch.randn(C, C, ks, device=dev) / math.sqrt(C * ks)
        wt = weights.pack_conv_f16s(w).to(dev)
        for dil in (1, 3, 5):
            pad = (ks - 1) * dil // 2
            pk = dict(pro=ops.PRO_ADAIN_SNAKE, stats=st, gamma=h[:, :C], beta=h[:, C:], alpha=alpha)
            for want in (False, True):
                _hooks.conv_path = "fused"
                t_f = timed(lambda: ops.conv1d(x, wt, C, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x, want_stats=want, **pk))

                def pair():
                    xs = ops.activate(x, **pk)
                    return ops.conv1d_xs(xs, wt, C, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x, want_stats=want)
                t_p = timed(pair)
                t_a = timed(lambda: ops.activate(x, **pk))
                print("C=%d L=%d k=%d dil=%d stats=%d: fused %.3f ms, pair %.3f ms (activation pass %.3f) -> %s" % (
                    C, L, ks, dil, want, t_f, t_p, t_a, "fused" if t_f <= t_p else "PAIR x%.2f" % (t_f / t_p)), flush=True)
