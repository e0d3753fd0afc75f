"""CPU experiment: how much waveform error does an f16 hi/lo split (3 MFMA products, fp32 accumulate) of the
conv operands add, compared with the fp32 round-off the oracle itself carries?  Everything runs through the
oracle's decoder; `F.conv1d` is swapped for an emulation of the split product.

    python tools/probe_split_precision.py [tag] [T]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F

from _util import decoder_kwargs, manifest, rms
from oracle import st2_oracle as O
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
from styletts2_amd.decoder import Decoder

tag = sys.argv[1] if len(sys.argv) > 1 else "ljspeech"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dc = manifest(tag)["config"]["decoder"]
dec = Decoder(**decoder_kwargs(dc)).eval()
synth.init_synthetic_(dec, 1)
sd = dec.state_dict()
asr, F0, N, s, noise = synth.decoder_inputs(1, T, 3)

real_conv1d = F.conv1d
LO_SCALE = 2048.0
MODE = {"m": "fp32", "min_cin": 32}


def split(t, kind):
    if kind == "f16":
        hi = t.half().float()
        lo = ((t - hi) * LO_SCALE).half().float() / LO_SCALE
        return [hi, lo]
    if kind == "bf16x3":
        a = t.bfloat16().float()
        b = (t - a).bfloat16().float()
        c = (t - a - b).bfloat16().float()
        return [a, b, c]
    raise ValueError(kind)


def conv_emul(x, w, b=None, **kw):
    m = MODE["m"]
    if m == "fp32" or x.shape[1] < MODE["min_cin"] or kw.get("stride", 1) != 1:
        return real_conv1d(x, w, b, **kw)
    if m == "f16x2":  # hi*hi + hi*lo + lo*hi
        xs, ws = split(x, "f16"), split(w, "f16")
        pairs = [(0, 0), (0, 1), (1, 0)]
    elif m == "f16x2_4":  # all four products
        xs, ws = split(x, "f16"), split(w, "f16")
        pairs = [(0, 0), (0, 1), (1, 0), (1, 1)]
    elif m == "bf16x3":
        xs, ws = split(x, "bf16x3"), split(w, "bf16x3")
        pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    elif m == "f16":
        xs, ws = [x.half().float()], [w.half().float()]
        pairs = [(0, 0)]
    y = None
    for i, j in reversed(pairs):  # small terms first
        t = real_conv1d(xs[i], ws[j], None, **kw)
        y = t if y is None else y + t
    if b is not None:
        y = y + b.reshape(1, -1, 1)
    return y


def run(mode, dtype=torch.float32, har=None):
    MODE["m"] = mode
    sdd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}
    taps = {}
    O.F.conv1d = conv_emul
    try:
        with torch.no_grad():
            out = O.decoder(sdd, dc, asr.to(dtype), F0.to(dtype), N.to(dtype), s.to(dtype), noise=noise.to(dtype),
                            har=None if har is None else har.to(dtype), taps=taps)
    finally:
        O.F.conv1d = real_conv1d
    return out, taps


ref32, t32 = run("fp32")
har = t32["har"]
ref64, t64 = run("fp32", torch.float64, har=har)
print("signal RMS %.4g; fp32 oracle vs fp64 truth: waveform RMS %.3g" % (rms(ref64), rms(ref32.double() - ref64)))
for mode in ("f16x2", "f16x2_4", "bf16x3", "f16"):
    o, tp = run(mode, har=har)
    line = "%-8s vs fp64: %.3g   vs fp32 oracle: %.3g  max %.3g |" % (mode, rms(o.double() - ref64), rms(o - ref32),
                                                                    (o - ref32).abs().max().item())
    for k in ("front", "stage0", "stage1"):
        if k in tp:
            e = (tp[k] - t32[k]).abs().max().item() / t32[k].abs().max().item()
            line += " %s %.2g" % (k, e)
    print(line, flush=True)
