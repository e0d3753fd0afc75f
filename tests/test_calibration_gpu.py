"""Per-layer operand scales of the split-f16 convs on the device (include/st2.h st2_calibrate, ABI v20).

The reference's convs and Linears are fp32 at every magnitude (Modules/istftnet.py:68-74, 376-377; Modules/diffusion/
modules.py:484-490).  The split-f16 convs match that only where both f16 halves of the scaled operand are normal numbers;
by rule the scale is 8 (normalised inputs) or 1, which is fp32-class for O(1) tensors and 5-500 x worse for a layer whose
input sits at 1e-2 ... 1e-4.  These tests build checkpoints that put un-normalised conv inputs there (GELU / LeakyReLU
outputs, generator stage outputs, small LayerNorm gains), calibrate through the product path and hold the engine to the SAME
tap bars the O(1) checkpoints are held to; the two-sided telemetry (st2_debug_headroom) must show the low end before and
after."""
import pytest
import torch

from _util import WAVE_RMS_TOL, decoder_kwargs, manifest, rms
from oracle import ops_ref as R
from oracle import st2_oracle as O
from styletts2_amd import models, ops, pipeline
from styletts2_amd.decoder import Decoder
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)

pytestmark = pytest.mark.gpu
DEV = "cuda"
KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]


@pytest.mark.parametrize("path", ["xs", "fused"])
def test_headroom_reports_both_ends(path, monkeypatch):
    """One conv whose input sits at 1e-3: by rule (x_scale = 1) the telemetry must say so -- most of the operand's energy in
    elements whose lo half is subnormal, an implied relative error two orders above fp32's -- and with the calibrated scale
    both figures drop to the format's floor while the top of the range stays 3 bits clear."""
    from styletts2_amd import _hooks, weights
    monkeypatch.setattr(_hooks, "conv_path", "xs" if path == "xs" else "fused")
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(2, 96, 700, generator=gen) * 1e-3).to(DEV)
    w = weights.pack_conv_f16s(torch.randn(80, 96, 3, generator=gen) / 17.0).to(DEV)
    rows = {}
    for name, xsc in (("rule", None), ("calibrated", ops.calibrated_x_scale(float(x.abs().max())))):
        with ops.headroom() as h:
            if path == "xs":
                ops.conv1d_xs(ops.activate(x, x_scale=xsc), w, 80, 3, pad_left=1)
            else:
                ops.conv1d(x, w, 80, 3, pad_left=1, x_scale=xsc)
        assert len(h.rows) == 1 and h.rows[0]["kind"] == ("act_split" if path == "xs" else "fused conv")
        assert h.rows[0]["site"] == -1, "a per-kernel call belongs to no engine site"
        rows[name] = h.rows[0]
    r, c = rows["rule"], rows["calibrated"]
    assert r["x_scale"] == 1.0 and r["frac"] < 1e-6
    assert r["sub_share"] > 0.99 and 3e-6 < r["rel_err"] < 1e-4, r
    assert 4096.0 <= c["max_abs"] < 8192.0 and c["frac"] < 0.126
    assert c["sub_share"] < 0.01 and c["rel_err"] < 1.5e-7, c


def _small_decoder(tag, f):
    dc = manifest(tag)["config"]["decoder"]
    dec = Decoder(**decoder_kwargs(dc)).eval()
    synth.init_trained_like_(dec, 1)
    # front output, harmonic-source branch, resblock branches and the up-sampling convs scaled down: every generator stage --
    # i.e. the inputs of ups[i] (LeakyReLU / Snake prologue) and conv_post -- carries O(f) ... O(10 f) values instead of O(1)
    synth.scale_params_(dec, {"decode.3.conv2.": f, "decode.3.conv1x1.": f, "generator.ups.": 0.1,
                              "generator.noise_convs.": f, ".convs2.": f})
    return dc, dec


@pytest.mark.parametrize("tag,f", [("ljspeech", 1e-2), ("ljspeech", 1e-3), ("libritts", 1e-3)])
def test_calibrated_decoder_meets_the_tap_bars_at_small_magnitudes(tag, f):
    dc, dec = _small_decoder(tag, f)
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    asr, F0, N, s, noise = synth.decoder_inputs(2, 24, 3)
    asr = asr * f  # asr_res and the encode block's 1x1 shortcut read it un-normalised
    to = {}
    with torch.no_grad():
        O.decoder(sd, dc, asr, F0, N, s, noise=noise, taps=to)
        har = to["har"] if dc["type"] == "istftnet" else None  # tap-point protocol: iSTFTNet phase input injected
        ref = O.decoder(sd, dc, asr, F0, N, s, noise=noise, har=har) if har is not None else O.decoder(sd, dc, asr, F0, N, s, noise=noise)
    dec = dec.to(DEV)
    a = [t.to(DEV) for t in (asr, F0, N, s)]
    kw = dict(noise=noise.to(DEV), har=None if har is None else har.to(DEV))

    def run(taps=None):
        return dec(*a, taps=taps, **kw)

    ops.status(clear=True)
    with ops.headroom() as before:
        run()
    rep = pipeline.calibrate(run)
    eng = dec._eng
    assert rep["sites_set"] > 60 and rep["clamped_last_pass"] == 0 and rep["passes"] <= 2
    table = eng.calibration()
    launched = [r for r in table if r["seen"] > 0]
    assert len(launched) == rep["sites_set"]
    # the largest operand of every calibrated layer sits 3 bits below the clamp after a normalising prologue, (4096, 8192], and 5
    # bits below where the operand is free-ranging (asr_res / shortcut / stage inputs), (1024, 2048] -- ABI 22
    bands = [r["seen"] * r["x_scale"] for r in launched]
    assert all(4096.0 <= v < 8192.0 or 1024.0 <= v < 2048.0 for v in bands), sorted(bands)[:5]
    assert any(v >= 4096.0 for v in bands) and any(v < 2048.0 for v in bands)
    with ops.headroom() as after:
        te = {}
        out = run(te)
    torch.cuda.synchronize()
    assert ops.status(clear=True) == 0
    worst_before = max(r["rel_err"] for r in before.rows)
    worst_after = max(r["rel_err"] for r in after.rows)
    assert worst_before > 1e-6, "the checkpoint is meant to sit at the low end by rule (%g)" % worst_before
    assert worst_after < 2e-7, "calibrated: every operand at the format's floor (%g)" % worst_after
    assert max(r["frac"] for r in after.rows) < 0.126
    assert all(r["site"] >= 0 for r in after.rows), "engine launches carry their conv site"
    # the module-level tap bars of the O(1) checkpoints (tests/test_decoder_gpu.py), on a checkpoint two / three decades down
    for k in ["encode", "front"] + ["stage%d" % i for i in range(len(dc["upsample_rates"]))]:
        e = (te[k].cpu() - to[k]).abs().max().item() / to[k].abs().max().item()
        assert e < 2e-5, "%s rel err %g" % (k, e)
    assert rms(out.cpu() - ref) < WAVE_RMS_TOL * max(1.0, rms(ref))
    assert rms(out.cpu() - ref) < 2e-5 * ref.abs().max().item(), "waveform relative to its own (small) scale"
    # a table is part of the engine: same inputs, same bits
    assert torch.equal(run(), out)
    # ... and a cleared table is the rule again, bit for bit
    eng.set_calibration(None)
    with ops.headroom() as h2:
        run()
    assert [r["x_scale"] for r in h2.rows] == [r["x_scale"] for r in before.rows]


def _whole_model(tag):
    man = manifest(tag)
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    for i, k in enumerate(KEYS):
        synth.init_trained_like_(model[k], 10 + i)
        model[k].eval()
    # un-normalised conv inputs two decades down: the FFN intermediates of the denoiser and of PL-BERT (GELU outputs), the
    # text encoder's LSTM input (last LayerNorm gain), the duration / prosody LSTM inputs (bert_encoder output), the generator
    synth.scale_params_(model.diffusion, {"feed_forward.0.": 1e-2})
    synth.scale_params_(model.bert, {"ffn.": 1e-2})
    synth.scale_params_(model.text_encoder, {"cnn.2.1.": 1e-2})
    synth.scale_params_(model.bert_encoder, {"": 1e-2})
    synth.scale_params_(model.decoder, {"decode.3.conv2.": 1e-2, "decode.3.conv1x1.": 1e-2, "generator.ups.": 0.1,
                                        "generator.noise_convs.": 1e-2, ".convs2.": 1e-2})
    sds = {k: {n: t.clone() for n, t in model[k].state_dict().items()} for k in KEYS}
    return man, model, sds


@pytest.mark.parametrize("tag", ["ljspeech", "libritts"])
def test_calibrated_text_to_waveform_on_a_small_magnitude_checkpoint(tag):
    """The whole product path (st2_front_forward -> st2_prosody_forward -> st2_decoder_forward) on a trained-like checkpoint
    whose denoiser, PL-BERT, text-encoder, prosody and generator intermediates sit two decades below O(1): calibrated through
    `pipeline.calibrate`, held to the tap bars of tests/test_pipeline_gpu.py; LibriTTS = StyleTransformer1d + HiFi-GAN."""
    man, model, sds = _whole_model(tag)
    g = torch.Generator().manual_seed(0)
    B, N, steps = 2, 13, 5
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    lengths = torch.LongTensor([N] * B)
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    dur = torch.full((B, N), 2, dtype=torch.long)
    sine_noise = torch.randn(B, 600 * 2 * N, 9, generator=g)
    ref_s = torch.randn(B, 256, generator=g) if man["config"]["multispeaker"] else None
    to, te = {}, {}
    ref = O.inference(sds, man["config"], man["plbert"], tokens, lengths, noise, step_noise, sine_noise,
                      diffusion_steps=steps, ref_s=ref_s, durations=dur, taps=to)
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    args = (model, sampler, tokens.to(DEV), lengths, noise.to(DEV))
    kw = dict(diffusion_steps=steps, ref_s=None if ref_s is None else ref_s.to(DEV), durations=dur,
              step_noise=step_noise.to(DEV), sine_noise=sine_noise.to(DEV))
    # the product path (one engine for the front, one for the decoder) AND the tap-point path (taps= makes every stage run on its
    # own per-module engine, built on first use): both sets of engines are alive after the first pass and get their tables
    rep = pipeline.calibrate(lambda: (pipeline.inference(*args, **kw), pipeline.inference(*args, taps={}, **kw)), max_passes=2)
    rep = pipeline.calibrate(lambda: (pipeline.inference(*args, **kw), pipeline.inference(*args, taps={}, **kw)))
    assert rep["clamped_last_pass"] == 0 and rep["sites_set"] > 100
    engs = pipeline.model_engines(model, torch.device(DEV, torch.cuda.current_device()))
    assert {"front", "decoder"} <= set(engs)
    assert all(any(r["x_scale"] > 0 for r in e.calibration()) for k, e in engs.items() if k != "style")
    ops.status(clear=True)
    with ops.headroom() as h:
        out = pipeline.inference(*args, **kw)          # the product path
        pipeline.inference(*args, taps=te, **kw)       # the same stages with tap points
    torch.cuda.synchronize()
    assert ops.status(clear=True) == 0
    assert max(r["frac"] for r in h.rows) < 0.126
    worst = sorted(h.rows, key=lambda r: -r["rel_err"])[:4]
    for k, tol in (("s_pred", 5e-5), ("asr", 5e-5), ("en", 1e-4), ("F0", 1e-4), ("N", 1e-4)):
        e = (te[k].cpu() - to[k]).abs().max().item() / max(to[k].abs().max().item(), 1e-6)
        assert e < tol, "%s rel err %g" % (k, e)
    # the telemetry after calibration: every operand at the format's floor -- EXCEPT operands that are mostly exact zeros plus
    # a few large entries (their few non-zero small elements carry no weight in any sum); reported if it ever fails
    bad = [r for r in h.rows if r["rel_err"] > 2e-7]
    assert not bad, "operands above the floor after calibration: %s" % [
        (r["index"], r["kind"], r["pro"], r["C"], r["L"], r["x_scale"], "%.3g" % r["max_abs"], "%.2e" % r["rel_err"],
         "%.2f" % r["sub_share"], r["site"]) for r in worst]
    ref_style = to["s_pred"][:, :128]
    if ref_s is not None:
        ref_style = 0.3 * ref_style + 0.7 * ref_s[:, :128]
    har = to["har"].to(DEV)
    wave = model.decoder(to["asr"].to(DEV), to["F0"].to(DEV), to["N"].to(DEV), ref_style.contiguous().to(DEV),
                         noise=sine_noise.to(DEV), har=har)
    assert rms(wave.cpu() - ref) < WAVE_RMS_TOL
    assert rms(wave.cpu() - ref) < 5e-5 * ref.abs().max().item()
    if man["config"]["decoder"]["type"] == "hifigan":
        assert rms(out.cpu() - ref) < WAVE_RMS_TOL
    # the table travels: exported, cleared, re-installed -> the same bits as before
    dev = torch.device(DEV, torch.cuda.current_device())
    state = pipeline.calibration_state(model, dev)
    a = pipeline.inference(*args, **kw)
    pipeline.load_calibration_state(model, dev, {k: [] for k in state})
    pipeline.load_calibration_state(model, dev, state)
    assert torch.equal(pipeline.inference(*args, **kw), a)
