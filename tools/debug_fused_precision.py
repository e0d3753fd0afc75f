#!/usr/bin/env python
"""One AdaIN + Snake conv on data shaped like a small-magnitude generator stage (per-channel offsets of 5e-3, variation 8e-4:
var << eps, rstd ~ 316), fused kernel vs xs pair vs the exact-fp32 kernel, all against an fp64 evaluation of the contract."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import ops_ref as R  # noqa: E402
from styletts2_amd import _hooks, ops, weights  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
for C, L, ks, dil, wscale, sig, alpha_dec in ((64, 7200, 7, 3, 1.0, 8e-4, 1.0), (64, 7200, 7, 1, 1e-3, 8e-4, 1.0), (32, 14400, 11, 5, 1.0, 8e-4, 1.0),
                                              (128, 2400, 7, 3, 1.0, 8e-4, 1.0), (64, 7200, 7, 3, 1.0, 1.0, 1.0), (64, 7200, 7, 3, 1.0, 8e-4, 0.0),
                                              (64, 7200, 7, 3, 1.0, 8e-2, 1.0)):
    B = 2
    x = torch.randn(B, C, 1, generator=g) * 5e-3 * (sig / 8e-4 if sig > 1e-2 else 1.0) + torch.randn(B, C, L, generator=g) * sig
    w = torch.randn(C, C, ks, generator=g) / math.sqrt(C * ks) * wscale
    bias = torch.randn(C, generator=g) * 0.02 * wscale
    h = torch.randn(B, 2 * C, generator=g) * 0.5
    alpha = 10.0 ** ((torch.rand(C, generator=g) * 2 - 1) * alpha_dec)
    res = x.clone()
    st = R.instnorm_stats(x)
    kw = dict(dil=dil, pad_left=(ks - 1) * dil // 2, bias=bias, pro=R.PRO_ADAIN_SNAKE, stats=st, gamma=h[:, :C], beta=h[:, C:], alpha=alpha,
              res=res)
    exact = R.conv1d(x.double(), weights.pack_conv(w).double(), C, ks, **{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)
                                                                              for k, v in kw.items()})
    kwg = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
    out = {}
    with _hooks.override(conv_path="fused"):
        out["fused f16s"] = ops.conv1d(x.to(dev), weights.pack_conv_f16s(w).to(dev), C, ks, **kwg)
        out["exact f32 "] = ops.conv1d(x.to(dev), weights.pack_conv(w).to(dev), C, ks, **kwg)
    with _hooks.override(conv_path="xs"):
        PRO = ("pro", "stats", "gamma", "beta", "alpha")
        xs = ops.activate(x.to(dev), **{k: v for k, v in kwg.items() if k in PRO})
        out["xs pair   "] = ops.conv1d_xs(xs, weights.pack_conv_f16s(w).to(dev), C, ks, **{k: v for k, v in kwg.items() if k not in PRO})
    torch.cuda.synchronize()
    ref32 = R.conv1d(x, weights.pack_conv(w), C, ks, **kw)
    out["ATen fp32 "] = ref32
    # error of the conv term alone (y - res - bias), relative to ITS maximum: the residual hides it otherwise
    conv_exact = exact - res.double() - bias.double().view(1, -1, 1)
    line = "C %3d L %5d k %2d d %d w x%g sigma %g alpha 10^+-%g:" % (C, L, ks, dil, wscale, sig, alpha_dec)
    for name, y in out.items():
        e = (y.detach().cpu().double() - exact).abs().max().item()
        line += "  %s %.2e (conv term %.2e)" % (name.strip(), e / exact.abs().max().item(), e / conv_exact.abs().max().item())
    print(line, flush=True)
print("status 0x%x" % ops.status(clear=True))
