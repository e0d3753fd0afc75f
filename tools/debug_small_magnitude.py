#!/usr/bin/env python
"""Where does the engine lose precision on a small-magnitude checkpoint?  Decoder taps vs the oracle for (a) the C++ plan by rule,
(b) the C++ plan calibrated, (c) the per-kernel Python plan with EXACT-fp32 MFMA convs (conv_precision = "f32": no split-f16 at all):
what (c) shares with (a) / (b) is everything that is not a conv operand -- InstanceNorm statistics, Snake, interleave, iSTFT."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _util import decoder_kwargs, manifest, rms  # noqa: E402
from benchdata import synth  # noqa: E402
from oracle import st2_oracle as O  # noqa: E402
from styletts2_amd import _hooks, ops, pipeline  # noqa: E402
from styletts2_amd.decoder import Decoder  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "libritts"
    f = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
    dc = manifest(tag)["config"]["decoder"]
    dec = Decoder(**decoder_kwargs(dc)).eval()
    synth.init_trained_like_(dec, 1)
    synth.scale_params_(dec, {"decode.3.conv2.": f, "decode.3.conv1x1.": f, "generator.ups.": 0.1, "generator.noise_convs.": f,
                              ".convs2.": f})
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    asr, F0, N, s, noise = synth.decoder_inputs(2, 24, 3)
    asr = asr * f
    to, t64 = {}, {}
    with torch.no_grad():
        O.decoder(sd, dc, asr, F0, N, s, noise=noise, taps=to)
        O.decoder({k: v.double() for k, v in sd.items()}, dc, asr.double(), F0.double(), N.double(), s.double(), noise=noise.double(),
                  har=to["har"].double(), taps=t64)
    har = to["har"].cuda()  # istftnet [B, n_fft + 2, M]; hifigan [B, 1, L]
    dec = dec.cuda()
    a = [t.cuda() for t in (asr, F0, N, s)]
    keys = ["encode", "front"] + ["stage%d" % i for i in range(len(dc["upsample_rates"]))]

    def errs(te, ref):
        return {k: (te[k].cpu().double() - ref[k].double()).abs().max().item() / ref[k].double().abs().max().item() for k in keys}

    def run(taps=None):
        return dec(*a, noise=noise.cuda(), har=har, taps=taps)
    res = {}
    te = {}
    run(te)
    res["engine by rule"] = errs(te, t64)
    pipeline.calibrate(run)
    te = {}
    run(te)
    res["engine calibrated"] = errs(te, t64)
    dec._eng.set_calibration(None)
    for name, kw in (("python plan, exact-fp32 convs", dict(conv_precision="f32")),
                     ("python plan, f16s (library routing)", dict()),
                     ("python plan, f16s fused everywhere", dict(conv_path="fused"))):
        with _hooks.override(plan="python", **kw):
            dec._pk = None
            te = {}
            run(te)
            res[name] = errs(te, t64)
            dec._pk = None
    res["oracle fp32 (ATen CPU)"] = errs(to, t64)
    print("%s decoder, un-normalised stages scaled by %g: max |x - fp64 oracle| / max |fp64 oracle|" % (tag, f))
    print("%-32s" % "" + "".join("%12s" % k for k in keys))
    for name, e in res.items():
        print("%-32s" % name + "".join("%12.2e" % e[k] for k in keys))
    print("status 0x%x" % ops.status(clear=True))


if __name__ == "__main__":
    main()
