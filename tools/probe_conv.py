"""GPU probe: times st2_conv1d on the vocoder's dominant shapes (HIP events on the launch stream)."""
import json
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import ops, weights

dev = "cuda"
B = int(os.environ.get("PROBE_B", "8"))
cases = [  # C, L, ks, dil
    (128, 48001, 3, 1), (128, 48001, 7, 3), (128, 48001, 11, 5), (128, 48001, 11, 1),
    (256, 8000, 3, 1), (256, 8000, 7, 1), (256, 8000, 11, 5),
    (64, 120000, 7, 3), (32, 240000, 11, 1),
]
KERNELS = os.environ.get("PROBE_KERNELS", "f32,f16s").split(",")
rows = []
for (Cc, L, ks, dil) in cases:
    x = torch.randn(B, Cc, L, device=dev)
    w = torch.randn(Cc, Cc, ks, device=dev) / math.sqrt(Cc * ks)
    wts = {"f32": weights.pack_conv(w), "f16s": weights.pack_conv_f16s(w).to(dev)}
    bias = torch.randn(Cc, device=dev)
    st = ops.instnorm_stats(x)
    h = torch.randn(B, 2 * Cc, device=dev) * 0.3
    alpha = torch.rand(Cc, device=dev) + 0.5
    out = torch.empty_like(x)
    for kern, pro in [(k, p) for k in KERNELS for p in (ops.PRO_NONE, ops.PRO_ADAIN_SNAKE)]:
        wt = wts[kern]
        kw = dict(dil=dil, pad_left=(ks - 1) * dil // 2, bias=bias, out=out, pro=pro)
        if pro == ops.PRO_ADAIN_SNAKE:
            kw.update(stats=st, gamma=h[:, :Cc], beta=h[:, Cc:], alpha=alpha, res=x)
        for _ in range(2):
            ops.conv1d(x, wt, Cc, ks, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            ops.conv1d(x, wt, Cc, ks, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        flop = 2.0 * B * Cc * Cc * ks * L
        rows.append(dict(kernel=kern, C=Cc, L=L, ks=ks, dil=dil, pro=pro, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 1)))
        print(rows[-1], flush=True)
    # stats kernel
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.instnorm_stats(x, out=st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(dict(stats_C=Cc, L=L, ms=ms, GBs=B * Cc * L * 4 / ms / 1e6), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/probe_conv.json", "w"), indent=1)
