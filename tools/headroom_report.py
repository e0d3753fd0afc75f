#!/usr/bin/env python
"""Where does a checkpoint put every split-f16 conv operand in the f16 range -- BOTH ends?  (include/st2.h st2_debug_headroom)

    python tools/headroom_report.py [--config ljspeech|libritts|libritts_istftnet] [--checkpoint PATH.pth]
                                    [--trained-like] [--small 1e-2] [--tokens 40] [--batch 2] [--all] [--json OUT.json]

The split-f16 convs carry every activated operand as f16 hi + lo of u = x_scale * pro(x).  Top end: |u| > 65504 is clamped
(ST2_STATUS_F16_RANGE).  Low end: below ~2^-3 the lo half is a subnormal f16 and the operand keeps an absolute error of 2^-25
instead of 2^-22 relative.  Snake alpha, weight-norm gains and the scale of every GELU / LeakyReLU output are free parameters
of a trained checkpoint (Modules/istftnet.py:27-62), so this runs the whole text -> waveform product path once by rule and
once after `pipeline.calibrate` and prints, per conv launch: max |u| / 65504, the relative RMS error the split adds to the
operand (fp32 storage itself: 3.4e-8) and the share of operand energy whose lo half is subnormal; then the calibration table.
Without --checkpoint the weights are seeded synthetic ones: plain (`init_synthetic_`), `--trained-like` (log-normal gains,
log-uniform Snake alpha) and / or `--small F` (FFN / LayerNorm / generator-stage scales multiplied by F: un-normalised conv
inputs F instead of O(1), `benchdata.synth.scale_params_`).  For a real checkpoint + its saved config use
tools/validate_checkpoint.py, which prints the same tables beside the per-tap errors against the oracle.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="ljspeech", choices=["ljspeech", "libritts", "libritts_istftnet"])
    ap.add_argument("--checkpoint", default=None, help="a reference-layout checkpoint (models.load_checkpoint)")
    ap.add_argument("--trained-like", action="store_true")
    ap.add_argument("--small", type=float, default=0.0, help="scale factor of the un-normalised conv inputs (e.g. 1e-2)")
    ap.add_argument("--tokens", type=int, default=40)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--all", action="store_true", help="print every launch, not the 12 worst of each table")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from benchdata import manifest, synth
    from styletts2_amd import models, ops, pipeline
    from validate_checkpoint import headroom_summary, print_headroom
    man = manifest(a.config)
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    if a.checkpoint:
        models.load_checkpoint(model, None, a.checkpoint, load_only_params=True, ignore_modules=[])
    else:
        for i, k in enumerate(KEYS):
            (synth.init_trained_like_ if a.trained_like else synth.init_synthetic_)(model[k], 10 + i)
    if a.small:
        f = a.small
        synth.scale_params_(model.diffusion, {"feed_forward.0.": f})
        synth.scale_params_(model.bert, {"ffn.": f})
        synth.scale_params_(model.text_encoder, {"cnn.2.1.": f})
        synth.scale_params_(model.bert_encoder, {"": f})
        synth.scale_params_(model.decoder, {"decode.3.conv2.": f, "decode.3.conv1x1.": f, "generator.ups.": 0.1,
                                            "generator.noise_convs.": f, ".convs2.": f})
    for k in KEYS:
        model[k].eval().to("cuda")
    g = torch.Generator().manual_seed(0)
    B, N = a.batch, a.tokens
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    dur = torch.full((B, N), 2, dtype=torch.long)  # equal frame counts: one decoder call
    noise = torch.randn(B, 1, 256, generator=g).cuda()
    ref_s = torch.randn(B, 256, generator=g).cuda() if man["config"]["multispeaker"] else None
    sampler = models.make_sampler(model)

    def run():
        return pipeline.inference(model, sampler, tokens.cuda(), torch.LongTensor([N] * B), noise, diffusion_steps=a.steps,
                                  ref_s=ref_s, durations=dur)

    limit = 10 ** 6 if a.all else 12
    ops.status(clear=True)
    with ops.headroom() as h0:
        run()
    st0 = ops.status(clear=True)
    print_headroom(h0.rows, "by rule (x_scale 8 after a normalising prologue, else 1)", limit=limit)
    print("   device status word: 0x%x%s" % (st0, " (F16_RANGE raised)" if st0 & 1 else ""))
    rep = pipeline.calibrate(run)
    with ops.headroom() as h1:
        run()
    st1 = ops.status(clear=True)
    print_headroom(h1.rows, "after pipeline.calibrate (%d sites, %d pass(es))" % (rep["sites_set"], rep["passes"]), limit=limit)
    print("   device status word: 0x%x" % st1)
    dev = torch.device("cuda", torch.cuda.current_device())
    table = {k: [r for r in e.calibration() if r["x_scale"] > 0] for k, e in pipeline.model_engines(model, dev).items()}
    print("-- calibration table: %s" % ", ".join("%s %d sites" % (k, len(v)) for k, v in table.items()))
    for k, rows in table.items():
        for r in (rows if a.all else sorted(rows, key=lambda r: -r["x_scale"])[:4]):
            print("   %-8s %-70s %4d->%-4d k%-2d  max|pro(x)| %-10.4g x_scale %g" % (k, r["name"], r["C_in"], r["C_out"], r["ks"],
                                                                                   r["seen"], r["x_scale"]))
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"config": a.config, "trained_like": a.trained_like, "small": a.small,
                       "by_rule": headroom_summary(h0.rows), "calibrated": headroom_summary(h1.rows),
                       "rows_by_rule": h0.rows, "rows_calibrated": h1.rows, "table": table}, f, indent=1, default=float)
    worst = max(r["frac"] for r in h1.rows)
    return 0 if worst < 1.0 and st1 == 0 else 2


if __name__ == "__main__":
    sys.exit(main())
