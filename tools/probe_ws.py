"""GPU probe: the warp-specialised build of the fused conv (st2_conv1d_f16s_ws.h) against the one-role build on the
vocoder's AdaIN + Snake layers -- bitwise comparison of the outputs (and of the InstanceNorm partial sums the epilogue
emits) and time per launch.
    python tools/probe_ws.py [lib.so]     (a measurement build from tools/build_ws_ablate.sh: times only, 4 cases)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _hooks, _lib

ABL = len(sys.argv) > 1
if ABL:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from styletts2_amd import ops, weights  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
lib = _lib.load()
# (B, C_in, C_out, L, ks, dil)
cases = [(32, 64, 64, 120000, 3, 1), (32, 64, 64, 120000, 7, 3), (32, 64, 64, 120000, 11, 5), (32, 64, 64, 120000, 11, 1),
         (32, 32, 32, 240000, 3, 1), (32, 32, 32, 240000, 7, 5), (32, 32, 32, 240000, 11, 3),
         (32, 128, 128, 48001, 3, 1), (32, 128, 128, 48001, 3, 5), (32, 128, 128, 40000, 3, 3),
         (32, 128, 128, 24000, 7, 1), (32, 128, 128, 24000, 11, 5), (32, 256, 256, 4000, 3, 1),
         (3, 40, 72, 30011, 7, 3), (2, 24, 24, 70001, 11, 5), (5, 100, 130, 9001, 3, 5)]  # ragged channels / lengths


def timed(fn, n=2 if os.environ.get("PROBE_WS_PMC") else 5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if ABL:
    cases = [cases[0], cases[2], cases[5], cases[7]]
    print("library:", _lib.LIB_PATH)
if os.environ.get("PROBE_WS_PMC"):  # under rocprofv3 --pmc: few launches of three layer shapes
    cases = [cases[0], cases[2], cases[7]]
_hooks.conv_path = "fused"
bad = 0
for (B, Ci, Co, L, ks, dil) in cases:
    pitch = (L + 31) // 32 * 32
    x = torch.randn(B, Ci, pitch, device=dev)[:, :, :L]
    w = torch.randn(Co, Ci, ks, device=dev) / math.sqrt(Ci * ks)
    wt = weights.pack_conv_f16s(w).to(dev)
    bias = torch.randn(Co, device=dev)
    st = ops.instnorm_stats(x)
    h = torch.randn(B, 2 * Ci, device=dev) * 0.3
    alpha = torch.rand(Ci, device=dev) + 0.5
    pad = (ks - 1) * dil // 2
    akw = dict(pro=ops.PRO_ADAIN_SNAKE, stats=st, gamma=h[:, :Ci], beta=h[:, Ci:], alpha=alpha)
    res = torch.randn(B, Co, pitch, device=dev)[:, :, :L] if Ci != Co else x
    outs, times = {}, {}
    for name, variant in (("one-role", 1), ("warp-spec", 2)):
        lib.st2_conv1d_f16s_set_variant(variant)
        out = torch.empty((B, Co, pitch), device=dev)[:, :, :L]
        fn = lambda: ops.conv1d(x, wt, Co, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=res, **akw)  # noqa: E731
        times[name] = timed(fn)
        outs[name] = out.clone()
    lib.st2_conv1d_f16s_set_variant(0)
    same = torch.equal(outs["one-role"], outs["warp-spec"])
    bad += not same and not ABL
    print("B=%d C=%d->%d L=%d k=%d dil=%d: one-role %.3f ms, warp-specialised %.3f ms (x%.2f), bitwise %s, status %d"
          % (B, Ci, Co, L, ks, dil, times["one-role"], times["warp-spec"], times["one-role"] / times["warp-spec"],
             "equal" if same else "DIFFERENT (max |d| %.3g)" % (outs["one-role"] - outs["warp-spec"]).abs().max().item(),
             ops.status()), flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
