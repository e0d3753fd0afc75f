#!/usr/bin/env python
"""First contact with a real checkpoint: does the engine reproduce the reference's arithmetic on THESE weights?

    python tools/validate_checkpoint.py PATH.pth CONFIG.yml [--plbert-config Utils/PLBERT/config.yml]
           [--tokens 40] [--batch 1] [--steps 5] [--ref-style] [--no-calibrate] [--backend gpu|cpu] [--json OUT.json]

What `models.py:696-713` (load_checkpoint) and the notebooks' loading cell (Demo/Inference_LJSpeech.ipynb:187-215) do with
a `.pth` + its saved training config, then one short utterance two ways on the same weights, tokens and replayed noise:

  * the oracle (oracle/st2_oracle.py: the CPU restatement of the reference path, fp32 ATen) -- the checker;
  * the engine through its C ABI: st2_front_forward -> st2_prosody_forward -> st2_decoder_forward (the product path);

and prints (1) per-tap errors (t_en, d, s_pred, durations, asr, F0, N, decoder taps, waveform RMS / mel-L1 against the two
`north_star` bars), (2) the two-sided operand table of every split-f16 conv launch (st2_debug_headroom: top of the f16
range used, implied relative error of the hi/lo split, share of operand energy with a subnormal lo half) before and after
`pipeline.calibrate`, (3) the per-site calibration table and (4) the sticky device status word.  Exit code 0 = every bar
met.  No real checkpoint exists offline: the tests drive this tool on the synthetic full-layout checkpoint of
tests/test_checkpoint_layout.py; `--backend cpu` runs the C++ plans on host memory through the tests' CPU contracts (no
telemetry there: the probes are device code).

This is test / bring-up infrastructure: it imports oracle/ and (for --backend cpu) tests/_cpu_backend.py; nothing in
styletts2_amd/ imports it.
"""
import argparse
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import yaml  # noqa: E402

HOT = ["bert", "bert_encoder", "predictor", "decoder", "text_encoder", "diffusion"]
BARS = {"t_en": 5e-5, "d": 1e-4, "s_pred": 5e-5,  # s_pred = the mixed style (ref | s), the front's output
        "asr": 5e-5, "F0": 1e-4, "N": 1e-4, "encode": 2e-5, "front": 2e-5}
WAVE_RMS_TOL, MEL_L1_TOL = 1e-4, 1e-3  # BASELINE.json north_star


def _rel(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _rms(x):
    return x.detach().cpu().double().pow(2).mean().sqrt().item()


def load_model(ckpt_path, cfg_path, plbert_cfg_path=None):
    """(model, model_params dict, plbert params dict, {key: state_dict with `module.` stripped}) -- the notebooks' loading."""
    from styletts2_amd import models
    from styletts2_amd.weights import strip_module_prefix
    cfg = yaml.safe_load(open(cfg_path))
    mp = cfg["model_params"] if "model_params" in cfg else cfg
    pl = dict(models.PLBERT_DEFAULTS)
    if plbert_cfg_path:
        pl.update(yaml.safe_load(open(plbert_cfg_path))["model_params"])
    model = models.build_model(models.recursive_munch(mp), None, None, models.load_plbert(pl))
    models.load_checkpoint(model, None, ckpt_path, load_only_params=True)
    raw = torch.load(ckpt_path, map_location="cpu")["net"]
    missing = [k for k in HOT if k not in raw]
    if missing:
        raise SystemExit("checkpoint has no 'net' entries for %s" % missing)
    sds = {k: dict(strip_module_prefix(raw[k])) for k in HOT}
    return model, mp, pl, sds


def headroom_summary(rows):
    """Worst layer at each end of the f16 range + the per-launch table."""
    if not rows:
        return None
    top = max(rows, key=lambda r: r["frac"])
    low = max(rows, key=lambda r: r["rel_err"])
    return {"launches": len(rows), "top": {k: top[k] for k in ("index", "kind", "pro", "C", "L", "x_scale", "max_abs", "frac")},
            "low": {k: low[k] for k in ("index", "kind", "pro", "C", "L", "x_scale", "rel_err", "sub_share")},
            "above_eighth": sum(r["frac"] > 0.125 for r in rows), "clamped": sum(r["frac"] >= 1.0 for r in rows),
            "rel_err_above_3e-7": sum(r["rel_err"] > 3e-7 for r in rows)}


def print_headroom(rows, title, out=sys.stdout, limit=12):
    print("-- split-f16 operand range, %s: %d conv launches" % (title, len(rows)), file=out)
    if not rows:
        return
    print("   %4s %-10s %-12s %5s %7s %9s %11s %9s %9s %6s" % ("#", "kind", "prologue", "C", "L", "x_scale", "max|u|", "of 65504",
                                                               "rel_err", "sub%"), file=out)
    worst = sorted(rows, key=lambda r: -max(r["frac"] / 0.125, r["rel_err"] / 3e-7))[:limit]
    for r in sorted(worst, key=lambda r: r["index"]):
        flag = " CLAMPED" if r["frac"] >= 1.0 else (" <3 octaves left" if r["frac"] > 0.125 else "")
        flag += " low end" if r["rel_err"] > 3e-7 else ""
        print("   %4d %-10s %-12s %5d %7d %9g %11.4g %9.2e %9.2e %6.1f%s" % (
            r["index"], r["kind"], r["pro"], r["C"], r["L"], r["x_scale"], r["max_abs"], r["frac"], r["rel_err"],
            100.0 * r["sub_share"], flag), file=out)
    s = headroom_summary(rows)
    print("   top of the range: launch %d at %.3g of 65504 (%d above 1/8, %d clamped); low end: launch %d rel_err %.2e "
          "(fp32 storage 3.4e-8; %d launches above 3e-7)" % (s["top"]["index"], s["top"]["frac"], s["above_eighth"], s["clamped"],
                                                               s["low"]["index"], s["low"]["rel_err"], s["rel_err_above_3e-7"]),
          file=out)


def validate(ckpt_path, cfg_path, plbert_cfg_path=None, n_tokens=40, batch=1, steps=5, ref_style=None, calibrate=True,
             backend="gpu", out=sys.stdout):
    from oracle import st2_oracle as O  # checker
    from oracle.mel_ref import mel_spectrogram_t
    from styletts2_amd import engine, models, ops, pipeline
    model, mp, pl, sds = load_model(ckpt_path, cfg_path, plbert_cfg_path)
    multi = bool(mp.get("multispeaker", False))
    if ref_style is None:
        ref_style = multi
    g = torch.Generator().manual_seed(0)
    B, N = batch, n_tokens
    tokens = torch.randint(1, int(mp.get("n_token", 178)), (B, N), generator=g)
    tokens[:, 0] = 0  # the notebooks prepend id 0
    lengths = torch.LongTensor([N] * B)
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    ref_s = torch.randn(B, 256, generator=g) if ref_style else None
    res = {"checkpoint": os.path.basename(ckpt_path), "multispeaker": multi, "decoder": mp["decoder"]["type"], "tokens": N,
           "batch": B, "steps": steps, "backend": backend, "sigma_data": model.diffusion.diffusion.sigma_data}
    print("== %s (%s, %s decoder, sigma_data %.4g) on the %s backend: %d x %d tokens, %d diffusion steps" % (
        res["checkpoint"], "multispeaker" if multi else "single speaker", res["decoder"], res["sigma_data"], backend, B, N, steps),
        file=out)

    # ---- oracle: predicted durations first (they decide the frame count), then the whole path with taps ---------------------
    durs = []
    with torch.no_grad():
        for b in range(B):  # one utterance per call, as the notebooks run it (O.front batches equal frame counts only)
            tb = {}
            O.front(sds, mp, pl, tokens[b:b + 1], lengths[b:b + 1], noise[b:b + 1], step_noise[:, b:b + 1], diffusion_steps=steps,
                    ref_s=None if ref_s is None else ref_s[b:b + 1], durations=None, taps=tb)
            durs.append(tb["durations"][0])
    dur = torch.stack(durs)
    if len(set(dur.sum(dim=1).tolist())) > 1:  # one decoder call per frame count: validate the first utterance's group
        keep = [b for b in range(B) if int(dur[b].sum()) == int(dur[0].sum())]
        tokens, lengths, noise, step_noise, dur = tokens[keep], lengths[keep], noise[keep], step_noise[:, keep], dur[keep]
        ref_s = None if ref_s is None else ref_s[keep]
        B = len(keep)
    T = int(dur[0].sum())
    sine_noise = torch.randn(B, 600 * T, 9, generator=g)
    to = {}
    with torch.no_grad():
        ref = O.inference(sds, mp, pl, tokens, lengths, noise, step_noise, sine_noise, diffusion_steps=steps, ref_s=ref_s,
                          durations=dur, taps=to)

    # ---- engine: the product path's three C-ABI calls ---------------------------------------------------------------------
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "gpu" else None
    ctx = contextlib.nullcontext()
    if backend == "cpu":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _cpu_backend import cpu_backend
        ctx = cpu_backend()
    mv = (lambda t: t.to(dev)) if dev is not None else (lambda t: t)
    sampler = models.make_sampler(model)
    table, sigma0 = sampler.step_table(steps)
    with ctx:
        eng = engine.build_model_engine(model, dev)

        def run(har=None, taps=None):
            f = eng.front_forward(mv(tokens), mv(noise), mv(step_noise), table, sigma0, ref_s=None if ref_s is None else mv(ref_s),
                                  tail=0 if multi else 5)
            asr, F0, Nn = eng.prosody_forward(f["d_cm"], f["t_en"], mv(dur), f["s"], T, shift=mp["decoder"]["type"] == "hifigan")
            wave = eng.decoder_forward(asr, F0, Nn, f["ref"], noise=mv(sine_noise), har=har, taps=taps)
            return f, asr, F0, Nn, wave

        before = after = None
        if backend == "gpu":
            ops.status(clear=True)
            with ops.headroom() as h:
                run()
            before = h.rows
            res["status_uncalibrated"] = ops.status(clear=True)
            print_headroom(before, "by rule (x_scale 8 after a normalising prologue, else 1)", out)
            if calibrate:
                rep = pipeline.calibrate(run, engines=[eng])
                res["calibration"] = {k: rep[k] for k in ("passes", "sites_set", "clamped_last_pass")}
                print("-- calibrated %d conv sites in %d pass(es)" % (rep["sites_set"], rep["passes"]), file=out)
        te = {}
        if backend == "gpu":
            with ops.headroom() as h:
                f, asr, F0, Nn, wave = run(taps=te)
            after = h.rows
            torch.cuda.synchronize()
            if calibrate:
                print_headroom(after, "calibrated", out)
        else:
            f, asr, F0, Nn, wave = run(taps=te)
        # iSTFTNet takes torch.angle of the harmonic STFT as an input (flips by 2 pi under 1e-8 changes: SURVEY 7.3-2): the
        # conv path is compared with the oracle's harmonic features injected; HiFi-GAN end to end
        ref_style_vec = to["s_pred"][:, :128]
        if ref_s is not None:
            ref_style_vec = 0.3 * ref_style_vec + 0.7 * ref_s[:, :128]
        har = mv(to["har"]) if mp["decoder"]["type"] == "istftnet" else None
        wave_h = eng.decoder_forward(mv(to["asr"]), mv(to["F0"]), mv(to["N"]), mv(ref_style_vec.contiguous()), noise=mv(sine_noise),
                                     har=har)
        status = ops.status(clear=True) if backend == "gpu" else 0

    # ---- report -----------------------------------------------------------------------------------------------------------
    got = {"t_en": f["t_en"], "d": f["d_cm"].transpose(1, 2), "s_pred": f["s_pred"], "asr": asr,
           "F0": F0, "N": Nn, "encode": te.get("encode"), "front": te.get("front")}
    want = {"t_en": to.get("t_en"), "d": to.get("d"), "s_pred": to["s_mixed"].reshape(B, -1), "asr": to["asr"], "F0": to["F0"],
            "N": to["N"], "encode": to.get("encode"), "front": to.get("front")}
    ok = True
    res["taps"] = {}
    print("-- taps (max |engine - oracle| / max |oracle|)", file=out)
    for k, bar in BARS.items():
        if got.get(k) is None or want.get(k) is None:
            continue
        a, b = got[k].detach().cpu(), want[k]
        if a.shape != b.shape:
            a = a.reshape(b.shape)
        e = _rel(a, b)
        res["taps"][k] = e
        ok &= e < bar
        print("   %-8s %.2e  (bar %.0e)%s" % (k, e, bar, "" if e < bar else "  <-- FAIL"), file=out)
    dur_e = f["durations"].detach().cpu()
    res["durations_equal"] = bool(torch.equal(dur_e, to["durations"])) if "durations" in to else None
    w_rms = _rms(wave_h.cpu() - ref)
    mel = (mel_spectrogram_t(wave_h.detach().cpu().float().reshape(-1, wave_h.shape[-1])) -
           mel_spectrogram_t(ref.float().reshape(-1, ref.shape[-1]))).abs().mean().item()
    res.update(wave_rms_err=w_rms, mel_l1=mel, wave_rms_ref=_rms(ref), status=status, finite=bool(torch.isfinite(wave).all()))
    ok &= w_rms < WAVE_RMS_TOL * max(1.0, res["wave_rms_ref"]) and mel < MEL_L1_TOL and res["finite"] and status == 0
    print("   waveform RMS err %.2e (bar %.0e, reference RMS %.3g), mel-L1 %.2e (bar %.0e), end-to-end output finite: %s" % (
        w_rms, WAVE_RMS_TOL, res["wave_rms_ref"], mel, MEL_L1_TOL, res["finite"]), file=out)
    print("   predicted durations (engine, on its own d): %s" % ("equal to the oracle's" if res["durations_equal"] else
                                                                 "DIFFER from the oracle's" if res["durations_equal"] is False else "n/a"),
          file=out)
    ok &= res["durations_equal"] is not False
    print("-- device status word: 0x%x%s" % (status, "" if status == 0 else "  <-- see include/st2.h ST2_STATUS_*"), file=out)
    res["headroom_by_rule"], res["headroom_final"] = headroom_summary(before or []), headroom_summary(after or [])
    if backend == "gpu" and calibrate:
        cal = [r for r in eng.calibration() if r["x_scale"] > 0]
        res["calibration"]["table"] = cal
        lo = sorted(cal, key=lambda r: -r["x_scale"])[:5]
        print("-- largest calibrated scales (smallest inputs): " + "; ".join("%s x%g (max %.3g)" % (r["name"], r["x_scale"], r["seen"])
                                                                             for r in lo), file=out)
    res["ok"] = bool(ok)
    print("== %s" % ("every bar met" if ok else "FAILED"), file=out)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("config")
    ap.add_argument("--plbert-config", default=None)
    ap.add_argument("--tokens", type=int, default=40)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--ref-style", action="store_true", help="random reference style vector (default for multispeaker models)")
    ap.add_argument("--no-calibrate", action="store_true")
    ap.add_argument("--backend", default="gpu", choices=["gpu", "cpu"])
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = validate(a.checkpoint, a.config, a.plbert_config, a.tokens, a.batch, a.steps, a.ref_style or None, not a.no_calibrate,
                   a.backend)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1, default=float)
    return 0 if res["ok"] else 2


if __name__ == "__main__":
    sys.exit(main())
