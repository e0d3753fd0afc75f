"""GPU probe: times the conv kernels on the vocoder's dominant shapes (HIP events on the launch stream).
fused = st2_conv1d_f16s (prologue inside the MFMA kernel); xs = st2_act_split + st2_conv1d_xs (both register
budgets, with and without the epilogue statistics)."""
import json
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _hooks, _lib, ops, weights

dev = "cuda"
B = int(os.environ.get("PROBE_B", "8"))
cases = [  # C, L, ks, dil
    (128, 48001, 3, 1), (128, 48001, 7, 3), (128, 48001, 11, 5), (128, 48001, 11, 1),
    (256, 8000, 3, 1), (256, 8000, 7, 1), (256, 8000, 11, 5),
    (64, 120000, 3, 1), (64, 120000, 7, 3), (64, 120000, 11, 5), (32, 240000, 3, 1), (32, 240000, 7, 1), (32, 240000, 11, 1),
    (128, 40000, 3, 1), (128, 40000, 7, 3), (1024, 400, 3, 1),
]
lib = _lib.load()


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


rows = []
for (Cc, L, ks, dil) in cases:
    pitch = (L + 31) // 32 * 32  # 128-byte aligned rows, as the C++ plan lays its workspace out
    x = torch.randn(B, Cc, pitch, device=dev)[:, :, :L]
    w = torch.randn(Cc, Cc, ks, device=dev) / math.sqrt(Cc * ks)
    wt = weights.pack_conv_f16s(w).to(dev)
    bias = torch.randn(Cc, device=dev)
    st = ops.instnorm_stats(x)
    h = torch.randn(B, 2 * Cc, device=dev) * 0.3
    alpha = torch.rand(Cc, device=dev) + 0.5
    out = torch.empty((B, Cc, pitch), device=dev)[:, :, :L]
    flop = 2.0 * B * Cc * Cc * ks * L
    pad = (ks - 1) * dil // 2
    akw = dict(pro=ops.PRO_ADAIN_SNAKE, stats=st, gamma=h[:, :Cc], beta=h[:, Cc:], alpha=alpha)
    _hooks.conv_path = "fused"
    r = dict(C=Cc, L=L, ks=ks, dil=dil)
    r["fused_pro0"] = timed(lambda: ops.conv1d(x, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out))
    r["fused_pro3"] = timed(lambda: ops.conv1d(x, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x, **akw))
    r["fused_res_stats"] = timed(lambda: ops.conv1d(x, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x,
                                                    want_stats=True, **akw))
    r["stats"] = timed(lambda: ops.instnorm_stats(x, out=st))
    r["act"] = timed(lambda: ops.activate(x, **akw))
    xs = ops.activate(x, **akw)
    r["xs_plain"] = timed(lambda: ops.conv1d_xs(xs, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out))
    r["xs_res"] = timed(lambda: ops.conv1d_xs(xs, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x))
    r["xs_res_stats"] = timed(lambda: ops.conv1d_xs(xs, wt, Cc, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x,
                                                    want_stats=True))
    r = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
    r["tflops_fused_pro3"] = round(flop / r["fused_pro3"] / 1e9, 1)
    r["tflops_xs_conv"] = round(flop / r["xs_plain"] / 1e9, 1)
    r["act_GBps"] = round(B * Cc * L * 8 / r["act"] / 1e6, 1)
    r["layer_fused_ms"] = round(r["fused_res_stats"], 4)           # prologue in the MFMA kernel, statistics from its epilogue
    r["layer_xs_ms"] = round(r["act"] + r["xs_res_stats"], 4)     # activation pass + conv on planes, same statistics
    rows.append(r)
    print(r, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/probe_conv.json", "w"), indent=1)
