"""Module-level C ABI on the GPU: `st2_decoder_forward` / `st2_sampler_run` -- the C++ launch
plans of csrc/st2_engine.hip -- against the per-kernel Python plans (decoder.py / diffusion.py, _hooks.override(plan="python")).  Both
issue the same kernels with the same arguments, so their results must be BITWISE equal; the Python plans are in turn
held to the oracle by test_decoder_gpu.py / test_sampler_gpu.py.  Also: a whole decoder call captured in a hipGraph
(the entry point allocates nothing and never synchronises), replayed on new inputs."""
import os

import pytest
import torch

from _util import decoder_kwargs, manifest
from styletts2_amd import _hooks, engine, models
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
from styletts2_amd.decoder import Decoder

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _plan(mode):
    return _hooks.override(plan=mode)


def _decoder(tag, wseed=1):
    dc = manifest(tag)["config"]["decoder"]
    dec = Decoder(**decoder_kwargs(dc)).eval()
    synth.init_synthetic_(dec, wseed)
    return dc, dec.to(DEV)


@pytest.mark.parametrize("tag,B,T", [("ljspeech", 2, 10), ("ljspeech", 1, 57), ("libritts", 2, 10), ("libritts", 1, 33),
                                     ("libritts_istftnet", 3, 16)])
def test_decoder_engine_equals_python_plan_bitwise(tag, B, T):
    dc, dec = _decoder(tag)
    asr, F0, N, s, noise = (t.to(DEV) for t in synth.decoder_inputs(B, T, 5))
    te, tp = {}, {}
    with _plan("engine"):
        assert engine.plan_mode() == "engine"
        out_e = dec(asr, F0, N, s, noise=noise, taps=te)
    with _plan("python"):
        out_p = dec(asr, F0, N, s, noise=noise, taps=tp)
    torch.cuda.synchronize()
    assert getattr(dec, "_eng", None) is not None, "the engine path did not run"
    assert out_e.shape == out_p.shape == (B, 1, 600 * T)
    for k in ("encode", "front", "har_source", "stage0", "stage1"):
        assert torch.equal(te[k].reshape(-1), tp[k].reshape(-1)), k
    assert torch.equal(out_e, out_p)
    # the harmonic-feature injection path of the tap-point protocol
    with _plan("engine"):
        out_e2 = dec(asr, F0, N, s, noise=noise, har=tp["har"])
    with _plan("python"):
        out_p2 = dec(asr, F0, N, s, noise=noise, har=tp["har"])
    assert torch.equal(out_e2, out_p2)


def _diffusion(tag, seed=2):
    man = manifest(tag)
    args = models.recursive_munch(man["config"])
    tcls = models.StyleTransformer1d if args.multispeaker else models.Transformer1d
    tr = tcls(channels=args.style_dim * 2, context_embedding_features=768, context_features=args.style_dim * 2,
              embedding_max_length=512, **args.diffusion.transformer)
    diff = models.AudioDiffusionConditional(tr, sigma_data=args.diffusion.dist.sigma_data).eval()
    synth.init_synthetic_(diff, seed)
    return man, diff.to(DEV)


@pytest.mark.parametrize("tag,B,N,steps,scale,ragged", [("ljspeech", 2, 37, 5, 1.0, False), ("ljspeech", 3, 100, 5, 1.5, False),
                                                        ("libritts", 2, 64, 10, 1.0, False), ("libritts", 1, 130, 5, 2.0, False),
                                                        ("ljspeech", 3, 48, 5, 1.0, True), ("libritts", 2, 80, 5, 1.5, True),
                                                        # B * N >= 256: the multispeaker q / kv convs as one token-merged
                                                        # GEMM with per-utterance affine rows (st2_act_split gb_seg)
                                                        ("libritts", 4, 80, 4, 1.5, False), ("libritts", 3, 96, 4, 1.0, True)])
def test_sampler_engine_equals_python_plan_bitwise(tag, B, N, steps, scale, ragged):
    man, diff = _diffusion(tag)
    sampler = models.DiffusionSampler(diff.diffusion, sampler=models.ADPM2Sampler(),
                                      sigma_schedule=models.KarrasSchedule(sigma_min=0.0001, sigma_max=3.0, rho=9.0),
                                      clamp=False)
    g = torch.Generator().manual_seed(N)
    noise = torch.randn(B, 1, 256, generator=g).to(DEV)
    emb = torch.randn(B, N, 768, generator=g).to(DEV)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g).to(DEV)
    kw = dict(embedding=emb, embedding_scale=scale, num_steps=steps, step_noise=step_noise)
    if man["config"]["multispeaker"]:
        kw["features"] = torch.randn(B, 256, generator=g).to(DEV)
    if ragged:
        kw["lengths"] = torch.tensor([N - 7 * b for b in range(B)], dtype=torch.int32, device=DEV)
    te, tp = {}, {}
    with _plan("engine"):
        out_e = sampler(noise, taps=te, **kw)
    with _plan("python"):
        out_p = sampler(noise, taps=tp, **kw)
    torch.cuda.synchronize()
    assert out_e.shape == out_p.shape == (B, 1, 256)
    assert set(te) == set(tp) and len(tp) == steps - 1
    for k in tp:
        assert torch.equal(te[k].reshape(-1), tp[k].reshape(-1)), k
    assert torch.equal(out_e, out_p)


@pytest.mark.parametrize("tag", ["ljspeech", "libritts"])
def test_decoder_forward_is_graph_capturable(tag):
    """st2_decoder_forward allocates nothing and never synchronises: after one eager call of the shape it is captured
    whole in a hipGraph (torch.cuda.CUDAGraph on a side stream) and the replay on NEW input values reproduces the eager
    result bitwise."""
    dc, dec = _decoder(tag)
    B, T = 2, 12
    a0 = [t.to(DEV) for t in synth.decoder_inputs(B, T, 7)]
    a1 = [t.to(DEV) for t in synth.decoder_inputs(B, T, 8)]
    with _plan("engine"):
        ref0 = dec(*a0[:4], noise=a0[4]).clone()
        ref1 = dec(*a1[:4], noise=a1[4]).clone()
        static = [t.clone() for t in a0]
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = dec(*static[:4], noise=static[4])
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref0)
        for dst, src in zip(static, a1):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref1)


@pytest.mark.parametrize("B,N,shift", [(2, 23, False), (3, 40, True)])
def test_prosody_engine_equals_python_plan_bitwise(B, N, shift):
    """st2_prosody_forward (C++ plan: alignment expansion + ProsodyPredictor.F0Ntrain) against the per-kernel Python
    plan (pipeline.expand_by_durations + predictor.F0Ntrain): same kernels, same arguments -> bitwise equal."""
    from styletts2_amd import pipeline
    from styletts2_amd.text import ProsodyPredictor
    pred = ProsodyPredictor(style_dim=128, d_hid=512, nlayers=3, max_dur=50).eval()
    synth.init_synthetic_(pred, 7)
    pred = pred.to(DEV)
    g = torch.Generator().manual_seed(B * 100 + N)
    d_cm = torch.randn(B, 640, N, generator=g).to(DEV)
    t_en = torch.randn(B, 512, N, generator=g).to(DEV)
    s = torch.randn(B, 128, generator=g).to(DEV)
    dur = torch.randint(1, 9, (B, N), generator=g)
    T = int(dur.sum(dim=1).max())
    for b in range(B):
        dur[b, -1] += T - int(dur[b].sum())
    dur = dur.to(DEV)
    en = pipeline.expand_by_durations(d_cm, dur, T, shift=shift)
    asr_p = pipeline.expand_by_durations(t_en, dur, T, shift=shift)
    f0_p, n_p = pred.F0Ntrain(en, s)
    eng = engine.build_predictor_engine(pred, torch.device(DEV))
    asr_e, f0_e, n_e = eng.prosody_forward(d_cm, t_en, dur, s, T, shift=shift)
    torch.cuda.synchronize()
    assert torch.equal(asr_e, asr_p)
    assert torch.equal(f0_e, f0_p) and torch.equal(n_e, n_p)


def _close(a, b, rel):
    scale = max(1.0, float(b.abs().max()))
    return float((a - b).abs().max()) <= rel * scale


@pytest.mark.parametrize("ragged", [False, True])
def test_text_plan_engine_equals_python_plan(ragged):
    """st2_text_forward (C++ plan) against text.TextEncoder.forward (per-kernel Python plan): the same conv / colnorm /
    LSTM kernels with the same arguments, the embedding gather + mask in `st2_embed_tokens` instead of torch indexing."""
    from styletts2_amd.text import TextEncoder
    enc = TextEncoder(channels=512, kernel_size=5, depth=3, n_symbols=178).eval()
    synth.init_synthetic_(enc, 9)
    enc = enc.to(DEV)
    B, N = 3, 37
    g = torch.Generator().manual_seed(6)
    tokens = torch.randint(1, 178, (B, N), generator=g)
    lengths = torch.tensor([N, N - 9, N - 2]) if ragged else torch.full((B,), N)
    mask = torch.arange(N).unsqueeze(0) >= lengths.unsqueeze(1)
    tokens = tokens.masked_fill(mask, 0).to(DEV)
    ref = enc(tokens, lengths, mask.to(DEV))
    eng = engine.build_text_engine(enc, torch.device(DEV))
    out = eng.text_forward(tokens, lengths.to(torch.int32).to(DEV) if ragged else None)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    print("text plan max |diff| = %.3e (bitwise: %s)" % (float((out - ref).abs().max()), torch.equal(out, ref)))
    assert torch.equal(out, ref)


@pytest.mark.parametrize("ragged,tail", [(False, 5), (True, 0)])
def test_duration_plan_engine_matches_python_plan(ragged, tail):
    """st2_duration_forward (C++ plan) against DurationEncoder.forward + pipeline.predict_durations.  The AdaLayerNorm
    style projections are a library kernel in the C++ plan and a torch GEMM in the Python plan, so d_cm is compared at
    fp32 rounding level; the integer durations must be identical."""
    from styletts2_amd import pipeline
    from styletts2_amd.text import ProsodyPredictor
    pred = ProsodyPredictor(style_dim=128, d_hid=512, nlayers=3, max_dur=50).eval()
    synth.init_synthetic_(pred, 7)
    pred = pred.to(DEV)
    B, N = 3, 29
    g = torch.Generator().manual_seed(17)
    d_en = torch.randn(B, 512, N, generator=g).to(DEV)
    s = torch.randn(B, 128, generator=g).to(DEV)
    lengths = torch.tensor([N, N - 7, N - 1]) if ragged else torch.full((B,), N)
    mask = torch.arange(N).unsqueeze(0) >= lengths.unsqueeze(1)
    d_p = pred.text_encoder(d_en, s, lengths, mask.to(DEV))                       # [B, N, 640]
    holder = type("M", (), {"predictor": pred})()
    dur_p = pipeline.predict_durations(holder, d_p, lj_tail=tail == 5, input_lengths=lengths)
    eng = engine.build_predictor_engine(pred, torch.device(DEV))
    d_e, dur_e = eng.duration_forward(d_en, s, lengths.to(torch.int32).to(DEV) if ragged else None, tail=tail)
    torch.cuda.synchronize()
    print("duration plan max |diff| = %.3e, durations equal: %s" % (float((d_e - d_p.transpose(1, 2)).abs().max()),
                                                                    torch.equal(dur_e, dur_p)))
    assert _close(d_e, d_p.transpose(1, 2), 2e-5)
    assert dur_e.dtype == torch.int64 and torch.equal(dur_e, dur_p)
    if ragged:
        assert int(dur_e[1, N - 7:].sum()) == 0


@pytest.mark.parametrize("ragged", [False, True])
def test_bert_plan_engine_equals_python_plan_bitwise(ragged):
    """st2_bert_forward (C++ plan) against CustomAlbert.forward_engine (per-kernel Python plan): same kernels, same
    arguments; the embedding sum comes from `st2_embed_tokens` (word + type, then + position: the Python plan's order)."""
    man = manifest("ljspeech")
    bert = models.load_plbert(man["plbert"]).eval()
    synth.init_synthetic_(bert, 15)
    bert = bert.to(DEV)
    B, N = 3, 41
    g = torch.Generator().manual_seed(12)
    tokens = torch.randint(1, 178, (B, N), generator=g)
    lengths = torch.tensor([N, N - 11, N - 1]) if ragged else torch.full((B,), N)
    mask = torch.arange(N).unsqueeze(0) >= lengths.unsqueeze(1)
    tokens = tokens.masked_fill(mask, 0).to(DEV)
    ref = bert.forward_engine(tokens, (~mask).int().to(DEV))
    eng = engine.build_bert_engine(bert, torch.device(DEV))
    out = eng.bert_forward(tokens, lengths.to(torch.int32).to(DEV))
    torch.cuda.synchronize()
    print("bert plan max |diff| = %.3e (bitwise: %s)" % (float((out - ref).abs().max()), torch.equal(out, ref)))
    assert torch.equal(out, ref)
    if not ragged:  # no lengths == every key valid
        assert torch.equal(eng.bert_forward(tokens, None), ref)


@pytest.mark.parametrize("tag,ragged,carry", [("ljspeech", True, True), ("libritts", False, False)])
def test_front_plan_engine_matches_python_front(tag, ragged, carry):
    """st2_front_forward (one C-ABI call) against pipeline._front_core (per-kernel Python plan + torch glue).  The two
    differ only where the Python front uses a torch GEMM / elementwise op (bert_encoder Linear, AdaLayerNorm style
    projections, the style mixing): compared at fp32 rounding level; the integer durations must be identical."""
    from styletts2_amd import pipeline
    man = manifest(tag)
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    for i, k in enumerate(["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]):
        synth.init_synthetic_(model[k], 10 + i)
        model[k].eval().to(DEV)
    B, N, steps = 3, 33, 4
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(1, 178, (B, N), generator=g)
    lengths = torch.LongTensor([N, N - 8, N - 1] if ragged else [N] * B)
    tokens = tokens.masked_fill(torch.arange(N).unsqueeze(0) >= lengths.unsqueeze(1), 0).to(DEV)
    noise = torch.randn(B, 1, 256, generator=g).to(DEV)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g).to(DEV)
    ref_s = torch.randn(B, 256, generator=g).to(DEV) if man["config"]["multispeaker"] else None
    s_prev = torch.randn(B, 256, generator=g).to(DEV) if carry else None
    sampler = models.make_sampler(model)
    kw = dict(diffusion_steps=steps, embedding_scale=1.5, alpha=0.3, beta=0.7, t=0.7, predict=True, lj_tail=tag == "ljspeech")
    lens_dev = lengths.to(torch.int32).to(DEV) if ragged else None
    outs = {}
    for mode in ("python", "engine"):
        with _hooks.override(plan=mode):
            outs[mode] = pipeline._front_core(model, sampler, tokens, lengths, lens_dev, noise, step_noise, ref_s, s_prev, **kw)
    torch.cuda.synchronize()
    p, e = outs["python"], outs["engine"]
    outs_plain_s = e["s"].clone()
    for k in ("t_en", "d", "s", "ref"):
        diff = float((p[k] - e[k]).abs().max())
        print("%s: max |diff| = %.3e of %.3e" % (k, diff, float(p[k].abs().max())))
    assert torch.equal(p["t_en"], e["t_en"])
    for k in ("d", "s", "ref"):
        assert _close(e[k], p[k], 5e-5), k
    assert torch.equal(p["durations"], e["durations"])
    # rows as consecutive sentences of one passage (st2_front_args.carry): the scan in the C++ plan against the scan of torch
    # ops in the Python front, and against the engine's own sentence-by-sentence calls handing the vector over
    first = None if s_prev is None else s_prev[:1]
    outs = {}
    for mode in ("python", "engine"):
        with _hooks.override(plan=mode):
            outs[mode] = pipeline._front_core(model, sampler, tokens, lengths, lens_dev, noise, step_noise, ref_s, first, carry=True,
                                              **kw)
    p, e = outs["python"], outs["engine"]
    for k in ("d", "s", "ref"):
        assert _close(e[k], p[k], 5e-5), k
    assert torch.equal(p["durations"], e["durations"])
    prev = first
    for b in range(B):
        one = pipeline._front_core(model, sampler, tokens[b:b + 1], lengths[b:b + 1], None if lens_dev is None else lens_dev[b:b + 1],
                                   noise[b:b + 1], step_noise[:, b:b + 1], None if ref_s is None else ref_s[b:b + 1], prev, **kw)
        prev = one["s_mixed"]
        assert _close(e["s_mixed"][b:b + 1], prev, 5e-5), b
        dd = (e["durations"][b:b + 1] - one["durations"]).abs()  # (a rounding tie of the duration head may flip one token)
        assert int(dd.max()) <= 1 and int((dd > 0).sum()) <= 1, b
    assert not _close(e["s"][1:], outs_plain_s[1:], 1e-3)  # the scan did mix rows 1.. with their predecessors
