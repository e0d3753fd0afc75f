"""GPU probe: the C = 32 / L = 240 000 / k = 3 HiFi-GAN conv through both builds of the fused kernel, in the variants the
vocoder issues (dilation 1 / 3 / 5, residual, residual + MRF accumulator + divide, InstanceNorm partial sums)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _hooks, _lib, ops, weights

dev = "cuda"
torch.manual_seed(0)
lib = _lib.load()
_hooks.conv_path = "fused"
B, C, L, ks = 32, 32, 240000, 3
pitch = (L + 31) // 32 * 32
x = torch.randn(B, C, pitch, device=dev)[:, :, :L]
r1 = torch.randn(B, C, pitch, device=dev)[:, :, :L]
r2 = torch.randn(B, C, pitch, device=dev)[:, :, :L]
w = torch.randn(C, C, ks, device=dev) / math.sqrt(C * ks)
wt = weights.pack_conv_f16s(w).to(dev)
bias = torch.randn(C, device=dev)
st = ops.instnorm_stats(x)
h = torch.randn(B, 2 * C, device=dev) * 0.3
alpha = torch.rand(C, device=dev) + 0.5
out = torch.empty((B, C, pitch), device=dev)[:, :, :L]


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


for dil in (1, 3, 5):
    for name, kw in (("plain", {}), ("stats", dict(want_stats=True)), ("res", dict(res=r1)), ("res+stats", dict(res=r1, want_stats=True)),
                     ("res+res2+div", dict(res=r1, res2=r2, div=3.0))):
        t = {}
        for vname, v in (("one-role", 1), ("ws", 2)):
            lib.st2_conv1d_f16s_set_variant(v)
            t[vname] = timed(lambda: ops.conv1d(x, wt, C, ks, dil=dil, pad_left=dil, bias=bias, out=out, pro=ops.PRO_ADAIN_SNAKE,
                                                stats=st, gamma=h[:, :C], beta=h[:, C:], alpha=alpha, **kw))
        lib.st2_conv1d_f16s_set_variant(0)
        print("dil=%d %-13s one-role %.3f ms, warp-specialised %.3f ms" % (dil, name, t["one-role"], t["ws"]), flush=True)
