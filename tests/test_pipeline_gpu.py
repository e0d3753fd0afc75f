"""End-to-end text -> waveform (rows a13-a17 + the whole chain): engine on the GPU vs the oracle's restatement of
the notebook `inference` cell, same weights, tokens and replayed noise."""
import pytest
import torch

from _util import WAVE_RMS_TOL, manifest, rms
from oracle import st2_oracle as O
from styletts2_amd import models, pipeline, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]


def _model(tag):
    man = manifest(tag)
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    for i, k in enumerate(KEYS):
        synth.init_synthetic_(model[k], 10 + i)
        model[k].eval()
    sds = {k: {n: t.clone() for n, t in model[k].state_dict().items()} for k in KEYS}
    return man, model, sds


@pytest.mark.parametrize("tag,ragged", [("ljspeech", False), ("libritts", False), ("ljspeech", True)])
def test_text_to_waveform_taps(tag, ragged):
    man, model, sds = _model(tag)
    g = torch.Generator().manual_seed(0)
    B, N, steps = 2, 13, 5
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    lengths = torch.LongTensor([N, N - 4] if ragged else [N] * B)
    if ragged:
        tokens[1, N - 4:] = 0
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    dur = torch.full((B, N), 2, dtype=torch.long)
    T = 2 * N
    sine_noise = torch.randn(B, 600 * T, 9, generator=g)
    ref_s = torch.randn(B, 256, generator=g) if man["config"]["multispeaker"] else None
    to, te = {}, {}
    ref = O.inference(sds, man["config"], man["plbert"], tokens, lengths, noise, step_noise, sine_noise,
                      diffusion_steps=steps, ref_s=ref_s, durations=dur, taps=to)
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    out = pipeline.inference(model, sampler, tokens.to(DEV), lengths, noise.to(DEV), diffusion_steps=steps,
                             ref_s=None if ref_s is None else ref_s.to(DEV), durations=dur,
                             step_noise=step_noise.to(DEV), sine_noise=sine_noise.to(DEV), taps=te)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    for k, tol in (("s_pred", 5e-5), ("asr", 5e-5), ("en", 1e-4), ("F0", 1e-4), ("N", 1e-4)):
        e = (te[k].cpu() - to[k]).abs().max().item() / max(to[k].abs().max().item(), 1e-6)
        assert e < tol, "%s rel err %g" % (k, e)
    # waveform at the 1e-4 bar: decoder fed with the oracle's own inputs and harmonic features (see test_decoder_gpu)
    ref_style = to["s_pred"][:, :128]
    if ref_s is not None:
        ref_style = 0.3 * ref_style + 0.7 * ref_s[:, :128]
    wave = model.decoder(to["asr"].to(DEV), to["F0"].to(DEV), to["N"].to(DEV), ref_style.contiguous().to(DEV),
                         noise=sine_noise.to(DEV), har=to["har"].to(DEV))
    assert rms(wave.cpu() - ref) < WAVE_RMS_TOL
    if man["config"]["decoder"]["type"] == "hifigan":  # no ill-conditioned STFT-phase input: true end-to-end bar
        assert rms(out.cpu() - ref) < WAVE_RMS_TOL


def test_predicted_durations_path_runs():
    man, model, sds = _model("ljspeech")
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(1, 178, (1, 17), generator=g)
    tokens[:, 0] = 0
    taps = {}
    out = pipeline.inference(model, sampler, tokens.to(DEV), diffusion_steps=3, taps=taps)
    T = int(taps["durations"].sum())
    assert out.shape == (1, 1, 600 * T) and bool(torch.isfinite(out).all())
    assert int(taps["durations"].min()) >= 1


@pytest.mark.parametrize("tag", ["libritts", "ljspeech"])
def test_long_form_streaming_matches_oracle_and_sequential(tag):
    """BASELINE.json configs[4]: pipeline.synthesize_long (two-stream front / decoder overlap, style carry-over) --
    (a) the overlapped run is bitwise the sequential run, (b) the style vector handed from sentence to sentence and the
    per-sentence HiFi-GAN waveforms match the oracle's restatement of the notebooks' long-form loop."""
    man, model, sds = _model(tag)
    g = torch.Generator().manual_seed(11)
    lens, steps = [9, 6, 12, 7], 3
    sentences = [torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, 178, (n - 1,), generator=g)]) for n in lens]
    noises = [torch.randn(1, 1, 256, generator=g) for _ in lens]
    step_noises = [torch.randn(steps - 1, 1, 1, 256, generator=g) for _ in lens]
    durs = [torch.full((1, n), 2, dtype=torch.long) for n in lens]
    sine = [torch.randn(1, 600 * 2 * n, 9, generator=g) for n in lens]
    multi = man["config"]["multispeaker"]
    ref_s = torch.randn(1, 256, generator=g) if multi else None
    ref_waves, s_prev = [], None
    for k in range(len(lens)):
        taps = {}
        w = O.inference(sds, man["config"], man["plbert"], sentences[k].reshape(1, -1), torch.LongTensor([lens[k]]),
                        noises[k], step_noises[k], sine[k], diffusion_steps=steps, ref_s=ref_s, durations=durs[k],
                        taps=taps, s_prev=s_prev, t=0.7, lj_tail=False)
        s_prev = taps["s_mixed"]
        ref_waves.append(w.reshape(-1)[:-100] if multi else w.reshape(-1))
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    d = lambda xs: [x.to(DEV) for x in xs]
    kw = dict(ref_s=None if ref_s is None else ref_s.to(DEV), t=0.7, diffusion_steps=steps, noises=d(noises),
              step_noises=d(step_noises), sine_noises=d(sine), durations=durs)
    order = []
    waves, style = pipeline.synthesize_long(model, sampler, d(sentences), overlap=True,
                                            on_chunk=lambda k, w: order.append(k), **kw)
    waves_seq, style_seq = pipeline.synthesize_long(model, sampler, d(sentences), overlap=False, **kw)
    torch.cuda.synchronize()
    assert order == list(range(len(lens)))
    assert torch.equal(style, style_seq) and all(torch.equal(a, b) for a, b in zip(waves, waves_seq))
    assert (style.cpu() - s_prev).abs().max().item() < 5e-5 * max(1.0, s_prev.abs().max().item())
    for w, r in zip(waves, ref_waves):
        assert w.shape == r.shape and bool(torch.isfinite(w).all())
        if man["config"]["decoder"]["type"] == "hifigan":
            assert rms(w.cpu() - r) < WAVE_RMS_TOL


def test_two_stream_inference_is_bitwise_the_single_stream_result():
    """pipeline.inference(front_stream=...): front on a side stream, decoder on the main stream, several batches
    back to back (the bench's overlap mode) == the single-stream results."""
    man, model, sds = _model("ljspeech")
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    g = torch.Generator().manual_seed(5)
    B, N, steps = 3, 21, 3
    batches = []
    for _ in range(3):
        tokens = torch.randint(1, 178, (B, N), generator=g)
        tokens[:, 0] = 0
        batches.append(dict(tokens=tokens.to(DEV), noise=torch.randn(B, 1, 256, generator=g).to(DEV),
                            step_noise=torch.randn(steps - 1, B, 1, 256, generator=g).to(DEV),
                            sine_noise=torch.randn(B, 600 * 2 * N, 9, generator=g).to(DEV)))
    dur = torch.full((B, N), 2, dtype=torch.long)

    def run(front):
        outs = []
        for b in batches:
            outs.append(pipeline.inference(model, sampler, b["tokens"], None, b["noise"], diffusion_steps=steps,
                                           durations=dur, step_noise=b["step_noise"], sine_noise=b["sine_noise"],
                                           front_stream=front))
        torch.cuda.synchronize()
        return outs

    ref = run(None)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    out = run(side)
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
