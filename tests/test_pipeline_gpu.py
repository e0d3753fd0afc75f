"""End-to-end text -> waveform (rows a13-a17 + the whole chain): engine on the GPU vs the oracle's restatement of
the notebook `inference` cell, same weights, tokens and replayed noise."""
import pytest
import torch

from _util import WAVE_RMS_TOL, manifest, rms
from oracle import st2_oracle as O
from styletts2_amd import models, pipeline
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)

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


@pytest.mark.parametrize("tag,ragged", [("ljspeech", False), ("libritts", False), ("ljspeech", True),
                                        ("libritts_istftnet", False), ("libritts_istftnet", True)])
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


@pytest.mark.parametrize("tag", ["ljspeech", "libritts", "libritts_istftnet"])
def test_predicted_durations_are_bit_exact(tag):
    """The path's integer output (row a14): duration LSTM -> Linear 512->50 -> sum of sigmoids -> round -> clamp(min=1)
    (Demo/Inference_LJSpeech.ipynb:296-301; +5 tail frames on the last phoneme in the LJSpeech flow only), B = 4
    utterances of N = 100 phonemes on the engine vs the oracle run one utterance at a time as the notebooks do:
    torch.equal, no tolerance.  The frame counts differ between utterances, so this also exercises the per-frame-count
    grouping of `prepare`."""
    man, model, sds = _model(tag)
    multi = man["config"]["multispeaker"]
    g = torch.Generator().manual_seed(101)
    B, N, steps = 4, 100, 5
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    lengths = torch.full((B,), N, dtype=torch.long)
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    ref_s = torch.randn(B, 256, generator=g) if multi else None
    want = []
    for b in range(B):
        to = {}
        with torch.no_grad():
            O.front(sds, man["config"], man["plbert"], tokens[b:b + 1], lengths[b:b + 1], noise[b:b + 1],
                    step_noise[:, b:b + 1], diffusion_steps=steps, ref_s=None if ref_s is None else ref_s[b:b + 1],
                    durations=None, taps=to)
        want.append(to["durations"][0])
    want = torch.stack(want)
    assert int(want.min()) >= 1
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    te = {}
    p = pipeline.prepare(model, sampler, tokens.to(DEV), lengths, noise.to(DEV), diffusion_steps=steps,
                         ref_s=None if ref_s is None else ref_s.to(DEV), step_noise=step_noise.to(DEV), taps=te,
                         allow_ragged=True)
    torch.cuda.synchronize()
    got = te["durations"].cpu()
    assert got.dtype == torch.int64 and torch.equal(got, want), (got - want).nonzero().tolist()
    frames = want.sum(dim=1).tolist()
    if len(set(frames)) > 1:
        assert sorted(b for idx, _ in p["groups"] for b in idx) == list(range(B))
        for idx, grp in p["groups"]:
            assert grp["asr"].shape == (len(idx), 512, frames[idx[0]]) and grp["F0"].shape[1] == 2 * frames[idx[0]]


def test_padded_batch_with_predicted_durations_equals_per_utterance_runs():
    """A right-padded batch through the predicted-duration path: the duration BiLSTM runs with each
    utterance's own length (its reverse pass must not start inside the padding), pad tokens get no frames, the +5
    tail lands on each utterance's own last phoneme, the style denoiser attends / averages over real tokens only, and
    every utterance is decoded at its own frame count.  Reference semantics = the notebooks' one-utterance-per-call:
    every row must equal the same utterance run alone, un-padded."""
    man, model, sds = _model("ljspeech")
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    g = torch.Generator().manual_seed(33)
    N, steps = 19, 3
    lens = [19, 15, 10]
    B = len(lens)
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    for b, n in enumerate(lens):
        tokens[b, n:] = 0
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    te = {}
    waves = pipeline.inference(model, sampler, tokens.to(DEV), torch.LongTensor(lens), noise.to(DEV),
                               diffusion_steps=steps, step_noise=step_noise.to(DEV), taps=te)
    torch.cuda.synchronize()
    assert isinstance(waves, list) and len(waves) == B
    for b, n in enumerate(lens):
        t1 = {}
        solo = pipeline.inference(model, sampler, tokens[b:b + 1, :n].to(DEV), torch.LongTensor([n]),
                                  noise[b:b + 1].to(DEV), diffusion_steps=steps,
                                  step_noise=step_noise[:, b:b + 1].to(DEV), taps=t1)
        d_b = te["durations"][b].cpu()
        assert torch.equal(d_b[:n], t1["durations"][0].cpu()) and int(d_b[n:].sum()) == 0
        assert int(d_b[n - 1]) > 5, "the LJSpeech tail belongs to the utterance's own last phoneme"
        assert (te["s_pred"][b] - t1["s_pred"][0]).abs().max().item() < 2e-5
        assert waves[b].shape[-1] == 600 * int(d_b.sum()) == solo.shape[-1]
        # un-injected iSTFTNet output: compare magnitudes loosely, the frame count and the front exactly
        assert bool(torch.isfinite(waves[b]).all())
    # the same batch against the oracle, one utterance at a time
    for b, n in enumerate(lens):
        to = {}
        with torch.no_grad():
            O.front(sds, man["config"], man["plbert"], tokens[b:b + 1, :n], torch.LongTensor([n]), noise[b:b + 1],
                    step_noise[:, b:b + 1], diffusion_steps=steps, durations=None, taps=to)
        assert torch.equal(te["durations"][b, :n].cpu(), to["durations"][0])
        assert (te["s_pred"][b].cpu() - to["s_pred"][0]).abs().max().item() < 5e-5


def test_golden_multispeaker_istftnet_front():
    """BASELINE.json configs[3] against vectors the REFERENCE modules produced (tests/golden/reference_vectors.npz
    `ms_*`, oracle/golden_vectors.py `frontend_vectors_multispeaker_istftnet`): multispeaker denoiser with `features`,
    style mixing, predicted durations without tail, un-shifted expansion, F0 / N."""
    import numpy as np
    from _util import GOLDEN
    import os
    gv = np.load(os.path.join(GOLDEN, "reference_vectors.npz"))
    man, model, sds = _model("libritts_istftnet")
    assert model.decoder.kind == "istftnet" and man["config"]["multispeaker"]
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    N, steps = 7, 3
    g = torch.Generator().manual_seed(21)
    tokens = torch.randint(1, 178, (1, N), generator=g)
    tokens[:, 0] = 0
    noise = torch.randn(1, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, 1, 1, 256, generator=g)
    ref_s = torch.randn(1, 256, generator=g)
    te = {}
    p = pipeline.prepare(model, sampler, tokens.to(DEV), torch.LongTensor([N]), noise.to(DEV), diffusion_steps=steps,
                         ref_s=ref_s.to(DEV), alpha=0.3, beta=0.7, step_noise=step_noise.to(DEV), taps=te)
    torch.cuda.synchronize()
    t = lambda k: torch.from_numpy(gv[k])
    assert torch.equal(te["durations"][0].cpu(), t("ms_dur").long())
    assert (te["s_pred"].cpu() - t("ms_s_pred")).abs().max().item() < 2e-5
    assert (p["ref"].cpu() - t("ms_ref")).abs().max().item() < 2e-5
    assert (te["F0"].cpu() - t("ms_F0")).abs().max().item() < 1e-4 * t("ms_F0").abs().max().item()
    assert (te["N"].cpu() - t("ms_N")).abs().max().item() < 1e-4 * t("ms_N").abs().max().item()
    assert (te["asr"].sum(dim=1).cpu() - t("ms_asr_sum")).abs().max().item() < 1e-3


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
    # front_batch: the sentences' fronts as ONE right-padded batch (0 = the whole passage, 3 = groups of three handing the
    # vector from call to call), style carry-over as a row scan: the sentence-by-sentence results, in sentence order
    got = {}
    for fb, ov, nds in ((0, True, 1), (0, False, 1), (3, True, 1), (0, True, 2), (2, True, 3)):
        order_f = []
        waves_f, style_f = pipeline.synthesize_long(model, sampler, d(sentences), overlap=ov, front_batch=fb, decode_streams=nds,
                                                    on_chunk=lambda k, w: order_f.append(k), **kw)
        torch.cuda.synchronize()
        got[(fb, ov) if nds == 1 else (fb, ov, nds)] = (waves_f, style_f)
        assert order_f == list(range(len(lens))), (fb, ov, order_f)
        assert (style_f - style).abs().max().item() < 5e-5 * max(1.0, style.abs().max().item()), (fb, ov)
        for w, r in zip(waves_f, ref_waves):
            assert w.shape == r.shape and bool(torch.isfinite(w).all())
            if man["config"]["decoder"]["type"] == "hifigan":
                assert rms(w.cpu() - r) < WAVE_RMS_TOL, (fb, ov)
    assert torch.equal(got[(0, True)][1], got[(0, False)][1])
    assert all(torch.equal(x, y) for x, y in zip(got[(0, True)][0], got[(0, False)][0]))  # overlapped == sequential, bitwise
    # independent sentences' decoders dealt onto two streams: the same waveforms, bit for bit
    assert all(torch.equal(x, y) for x, y in zip(got[(0, True, 2)][0], got[(0, False)][0]))


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


@pytest.mark.parametrize("tag,predict", [("libritts", False), ("ljspeech", True)])
def test_graphed_front_is_bitwise_the_eager_front(tag, predict):
    """pipeline.GraphedFront: the device-only front of a sentence (text encoder, PL-BERT, sampler, style mixing,
    duration encoder, optionally the duration head) replayed from ONE hipGraph per bucket == issued kernel by kernel,
    through the bucketed long-form loop (right-padded token rows, style carry-over, both stream modes)."""
    man, model, sds = _model(tag)
    g = torch.Generator().manual_seed(13)
    lens, steps = [9, 14, 12, 7, 16], 3
    sentences = [torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, 178, (n - 1,), generator=g)]) for n in lens]
    noises = [torch.randn(1, 1, 256, generator=g) for _ in lens]
    step_noises = [torch.randn(steps - 1, 1, 1, 256, generator=g) for _ in lens]
    durs = None if predict else [torch.full((1, n), 2, dtype=torch.long) for n in lens]
    multi = man["config"]["multispeaker"]
    ref_s = torch.randn(1, 256, generator=g).to(DEV) if multi else None
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model, graph=True)
    front = pipeline.GraphedFront(model, sampler)
    d = lambda xs: [x.to(DEV) for x in xs]
    sine = None if predict else d([torch.randn(1, 600 * 2 * n, 9, generator=g) for n in lens])
    kw = dict(ref_s=ref_s, t=0.7, diffusion_steps=steps, noises=d(noises), step_noises=d(step_noises),
              sine_noises=sine, durations=durs, bucket=8)
    if predict:  # the decoder draws its own SineGen noise per call: compare the style chain and the shapes only
        torch.manual_seed(0)
    w_e, s_e = pipeline.synthesize_long(model, sampler, d(sentences), overlap=False, **kw)
    if predict:
        torch.manual_seed(0)
    w_g, s_g = pipeline.synthesize_long(model, sampler, d(sentences), overlap=False, front=front, **kw)
    w_o, s_o = pipeline.synthesize_long(model, sampler, d(sentences), overlap=True, front=front, **kw)
    torch.cuda.synchronize()
    # one graph per signature: (bucket 8 | 16) x (first sentence: no carried style) x (exactly 16 tokens: no padding)
    assert 2 <= len(front._graphs) <= 4, len(front._graphs)
    assert torch.equal(s_e, s_g) and torch.equal(s_e, s_o)
    assert [w.shape for w in w_e] == [w.shape for w in w_g] == [w.shape for w in w_o]
    if not predict:
        assert all(torch.equal(a, b) for a, b in zip(w_e, w_g)) and all(torch.equal(a, b) for a, b in zip(w_e, w_o))
    assert all(bool(torch.isfinite(w).all()) for w in w_g)
    # a weight reload rebuilds the packed caches the recorded graphs point at: they must be dropped, not replayed
    old_graphs = dict(front._graphs)
    model.text_encoder.load_state_dict({k: v.clone() for k, v in model.text_encoder.state_dict().items()})
    if predict:
        torch.manual_seed(0)
    w_r, s_r = pipeline.synthesize_long(model, sampler, d(sentences), overlap=False, front=front, **kw)
    torch.cuda.synchronize()
    assert torch.equal(s_r, s_e)
    assert all(front._graphs[k] is not old_graphs.get(k) for k in front._graphs)
    # the whole passage through one graphed front call (one more signature: batch of 5, carried rows)
    if predict:
        torch.manual_seed(0)
    n_graphs = len(front._graphs)
    w_b, s_b = pipeline.synthesize_long(model, sampler, d(sentences), overlap=True, front=front, front_batch=0, **kw)
    w_b2, s_b2 = pipeline.synthesize_long(model, sampler, d(sentences), overlap=True, front=front, front_batch=0, **kw)
    torch.cuda.synchronize()
    assert len(front._graphs) == n_graphs + 1
    assert (s_b - s_e).abs().max().item() < 5e-5 * max(1.0, s_e.abs().max().item()) and torch.equal(s_b, s_b2)
    assert [w.shape for w in w_b] == [w.shape for w in w_e]
    if not predict:
        assert all(torch.equal(x, y) for x, y in zip(w_b, w_b2))


def test_real_validation_text_ragged_batch_matches_oracle_and_solo_runs():
    """The reference's own validation inputs (Data/val_list.txt column 2 -> TextCleaner; benchdata/val_phonemes_32.txt) instead of
    uniform synthetic rows: six real utterances of very different lengths as one right-padded batch, forced durations as in the
    bench's `ljspeech_ragged` leg.  One utterance is held to the oracle (front taps at their bars, the waveform at the 1e-4 bar
    with the oracle's harmonic features injected -- the iSTFTNet tap-point protocol); every row of the batch must be the same
    utterance synthesised alone, and the batch must return one waveform per utterance at its own length."""
    import bench
    man, model, sds = _model("ljspeech")
    tokens, lengths, noise, dur, lens = bench.ragged_inputs("cpu")
    pick = [21, 29, 3, 9, 0, 5]  # 50, 47, 84, 73, 131, 182 tokens
    tokens, lengths, noise, dur = tokens[pick], lengths[pick], noise[pick], dur[pick]
    lens = [lens[i] for i in pick]
    assert lens == [50, 47, 84, 73, 131, 182] and int(tokens.max()) < 178 and bool((tokens[:, 0] == 0).all())
    N = max(lens)
    tokens, dur = tokens[:, :N], dur[:, :N]
    g = torch.Generator().manual_seed(8)
    steps, B = 5, len(pick)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    sine_noise = torch.randn(B, 600 * 4 * N, 9, generator=g)
    b0, n0 = 1, lens[1]  # the 47-token utterance on the CPU oracle
    to = {}
    ref = O.inference(sds, man["config"], man["plbert"], tokens[b0:b0 + 1, :n0], lengths[b0:b0 + 1], noise[b0:b0 + 1],
                      step_noise[:, b0:b0 + 1], sine_noise[b0:b0 + 1, :600 * 4 * n0], diffusion_steps=steps, durations=dur[b0:b0 + 1, :n0],
                      taps=to)
    for k in KEYS:
        model[k].to(DEV)
    sampler = models.make_sampler(model)
    te = {}
    waves = pipeline.inference(model, sampler, tokens.to(DEV), lengths, noise.to(DEV), diffusion_steps=steps, durations=dur.to(DEV),
                               step_noise=step_noise.to(DEV), sine_noise=sine_noise.to(DEV), taps=te)
    torch.cuda.synchronize()
    assert isinstance(waves, list) and [w.shape[-1] for w in waves] == [600 * 4 * n for n in lens]
    assert all(bool(torch.isfinite(w).all()) for w in waves)
    # the per-utterance decoder calls dealt onto two streams (inference(decode_streams=)): bit for bit the sequential result
    from styletts2_amd import ops
    dev = torch.device(DEV, torch.cuda.current_device())
    dealt = pipeline.inference(model, sampler, tokens.to(DEV), lengths, noise.to(DEV), diffusion_steps=steps, durations=dur.to(DEV),
                               step_noise=step_noise.to(DEV), sine_noise=sine_noise.to(DEV),
                               decode_streams=[ops.aux_stream(dev, 0, index=1), ops.aux_stream(dev, 0, index=2)])
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(dealt, waves))
    e = (te["s_pred"][b0:b0 + 1].cpu() - to["s_pred"]).abs().max().item() / to["s_pred"].abs().max().item()
    assert e < 5e-5, "s_pred of the real utterance: %g" % e
    for b, n in enumerate(lens):  # every row == the utterance alone, un-padded
        t1 = {}
        solo = pipeline.inference(model, sampler, tokens[b:b + 1, :n].to(DEV), lengths[b:b + 1], noise[b:b + 1].to(DEV),
                                  diffusion_steps=steps, durations=dur[b:b + 1, :n].to(DEV), step_noise=step_noise[:, b:b + 1].to(DEV),
                                  sine_noise=sine_noise[b:b + 1, :600 * 4 * n].to(DEV), taps=t1)
        assert solo.shape[-1] == waves[b].shape[-1]
        assert (te["s_pred"][b] - t1["s_pred"][0]).abs().max().item() < 2e-5
        if b == b0:
            for k, tol in (("asr", 5e-5), ("en", 1e-4), ("F0", 1e-4), ("N", 1e-4)):
                err = (t1[k].cpu() - to[k]).abs().max().item() / max(to[k].abs().max().item(), 1e-6)
                assert err < tol, "%s rel err %g on real text" % (k, err)
            # waveform at the 1e-4 bar: the decoder on the oracle's inputs and harmonic features (tap-point protocol)
            wav = model.decoder(to["asr"].to(DEV), to["F0"].to(DEV), to["N"].to(DEV), to["s_pred"][:, :128].to(DEV),
                                noise=sine_noise[b0:b0 + 1, :600 * 4 * n0].to(DEV), har=to["har"].to(DEV))
            assert rms(wav.cpu() - ref) < WAVE_RMS_TOL
