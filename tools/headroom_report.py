#!/usr/bin/env python
"""How much of the f16 range does a checkpoint's decoder / vocoder use?  (include/st2.h st2_debug_headroom)

    python tools/headroom_report.py [--config ljspeech|libritts|libritts_istftnet] [--checkpoint PATH.pth]
                                    [--trained-like] [--frames 100] [--batch 2]

The split-f16 convs carry every activated operand as f16 hi + lo of x_scale * pro(x) and clamp at +-65504
(ST2_STATUS_F16_RANGE).  Weights are scaled per output row and cannot overflow; the activations can -- Snake alpha and the
weight-norm gains of a trained checkpoint are free parameters (Modules/istftnet.py:27-62).  This prints, per conv launch
of one decoder call, max |x_scale * pro(x)| / 65504 and flags every layer above 1/8 of the range (three octaves of
headroom left).  Without --checkpoint the weights are seeded synthetic ones: plain (`init_synthetic_`) or, with
--trained-like, with log-normal gains and log-uniform Snake alpha (`benchdata.synth.init_trained_like_`).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WARN_FRAC = 1.0 / 8


def report(rows, out=sys.stdout):
    worst = sorted(rows, key=lambda r: -r["frac"])
    print("%4s  %-10s %-12s %3s %5s %7s %7s %12s %9s" % ("#", "kind", "prologue", "B", "C", "L", "scale", "max|operand|", "of 65504"),
          file=out)
    for r in rows:
        flag = "  <-- CLAMPED" if r["frac"] >= 1.0 else ("  <-- < 3 octaves left" if r["frac"] > WARN_FRAC else "")
        print("%4d  %-10s %-12s %3d %5d %7d %7.0f %12.4g %9.2e%s" % (r["index"], r["kind"], r["pro"], r["B"], r["C"], r["L"],
                                                                    r["x_scale"], r["max_abs"], r["frac"], flag), file=out)
    if worst:
        w = worst[0]
        print("worst: launch %d (%s, %s, C=%d, L=%d) at %.3g of the f16 range; %d of %d launches above 1/8" % (
            w["index"], w["kind"], w["pro"], w["C"], w["L"], w["frac"], sum(r["frac"] > WARN_FRAC for r in rows), len(rows)),
            file=out)
    return worst[0]["frac"] if worst else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="ljspeech", choices=["ljspeech", "libritts", "libritts_istftnet"])
    ap.add_argument("--checkpoint", default=None, help="a reference-layout checkpoint (models.load_checkpoint)")
    ap.add_argument("--trained-like", action="store_true")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    from benchdata import manifest, synth
    from styletts2_amd import models, ops
    man = manifest(a.config)
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    if a.checkpoint:
        models.load_checkpoint(model, None, a.checkpoint, load_only_params=True, ignore_modules=[])
    else:
        (synth.init_trained_like_ if a.trained_like else synth.init_synthetic_)(model["decoder"], 1)
    dec = model["decoder"].eval().to("cuda")
    asr, F0, N, s, noise = synth.decoder_inputs(a.batch, a.frames, 3)
    with ops.headroom() as h:
        dec(asr.cuda(), F0.cuda(), N.cuda(), s.cuda(), noise=noise.cuda())
    worst = report(h.rows)
    st = ops.status(clear=True)
    print("device status word after the call: 0x%x%s" % (st, " (F16_RANGE raised)" if st & 1 else ""))
    return 0 if worst < 1.0 else 2


if __name__ == "__main__":
    sys.exit(main())
