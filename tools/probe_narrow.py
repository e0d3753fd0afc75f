"""GPU probe: the fused conv (st2_conv1d_f16s) on the HBM / VALU-bound layer shapes -- the narrow HiFi-GAN stages and the
k = 3 resblock convs -- with the library's tiles (128-column wave tiles, 2 workgroups / CU) against an experimental build
with 64-column wave tiles at 3 workgroups / CU (tools/bin/libst2_hip_narrow.so, built with -DST2_F16S_NARROW).
    python tools/probe_narrow.py [lib.so]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _hooks, _lib

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from styletts2_amd import ops, weights  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
B = 32
cases = [(64, 120000, 3, 1), (64, 120000, 7, 3), (64, 120000, 11, 5), (32, 240000, 3, 1), (32, 240000, 7, 1), (32, 240000, 11, 1),
         (128, 48001, 3, 1), (128, 40000, 3, 1)]


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


_hooks.conv_path = "fused"
print("library:", _lib.LIB_PATH)
for (Cc, L, ks, dil) in cases:
    pitch = (L + 31) // 32 * 32
    x = torch.randn(B, Cc, pitch, device=dev)[:, :, :L]
    w = torch.randn(Cc, Cc, ks, device=dev) / math.sqrt(Cc * ks)
    wt = weights.pack_conv_f16s(w).to(dev)
    bias = torch.randn(Cc, device=dev)
    st = ops.instnorm_stats(x)
    h = torch.randn(B, 2 * Cc, device=dev) * 0.3
    alpha = torch.rand(Cc, device=dev) + 0.5
    out = torch.empty((B, Cc, pitch), device=dev)[:, :, :L]
    pad = (ks - 1) * dil // 2
    akw = dict(pro=ops.PRO_ADAIN_SNAKE, stats=st, gamma=h[:, :Cc], beta=h[:, Cc:], alpha=alpha)
    t_res = timed(lambda: ops.conv1d(x, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x, **akw))
    t_plain = timed(lambda: ops.conv1d(x, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out, **akw))
    nbytes = B * Cc * L * 4 * 3
    print("C=%d L=%d k=%d dil=%d: fused + residual %.3f ms (%.2f TB/s of 12 B / element), without residual %.3f ms, checksum %.6e"
          % (Cc, L, ks, dil, t_res, nbytes / t_res / 1e9, t_plain, float(out.double().sum())), flush=True)
