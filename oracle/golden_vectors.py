"""Golden tap-point vectors produced by the UNMODIFIED reference modules (build container only).

Weights and inputs are regenerated from seeds on the test side (styletts2_amd/synth.py), so only reference
OUTPUTS are stored.  Run:  python -m oracle.make_golden vectors
"""
import os

import numpy as np
import torch

import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root: benchdata/, oracle/
from benchdata import synth  # noqa: E402  seeded synthetic weights / inputs (test + bench helper, not product code)
from oracle import ref_harness as RH  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]
DEC_B, DEC_T = 1, 8
SAMP = dict(B=2, N=21, steps=5)
E2E = dict(B=1, N=12, steps=3)


class replay_randn_like:
    """Feeds recorded tensors to the reference's in-forward torch.randn_like draws, matched by shape."""

    def __init__(self, by_shape):
        self.by_shape = {tuple(k): iter(v) for k, v in by_shape.items()}

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: (next(self.by_shape[tuple(x.shape)]).clone()
                                               if tuple(x.shape) in self.by_shape else self.orig(x, *a, **k))

    def __exit__(self, *a):
        torch.randn_like = self.orig


def decoder_vectors(tag, cfgname):
    model, args, cfg = RH.build_reference_model(cfgname)
    dec = model["decoder"]
    synth.init_synthetic_(dec, 1)
    asr, F0, N, s, noise = synth.decoder_inputs(DEC_B, DEC_T, 3)
    taps = {}
    dec.decode[3].register_forward_hook(lambda m, i, o: taps.__setitem__("front", o))
    dec.generator.m_source.register_forward_hook(
        lambda m, i, o: taps.__setitem__("har_source", o[0].transpose(1, 2).squeeze(1)))
    if cfg["model_params"]["decoder"]["type"] == "istftnet":
        orig = dec.generator.stft.transform

        def tr(x):
            a, b = orig(x)
            taps["har"] = torch.cat([a, b], 1)
            return a, b
        dec.generator.stft.transform = tr
    with torch.no_grad(), replay_randn_like({noise.shape: [noise]}):
        wave = dec(asr, F0, N, s)
    out = {"dec_%s_%s" % (tag, k): v.numpy() for k, v in taps.items()}
    out["dec_%s_wave" % tag] = wave.numpy()
    return out


def sampler_vectors(tag, cfgname):
    model, args, cfg = RH.build_reference_model(cfgname)
    ref = RH.load_reference()
    diff = model["diffusion"]
    synth.init_synthetic_(diff, 2)
    sampler = ref.sampler.DiffusionSampler(diff.diffusion, sampler=ref.sampler.ADPM2Sampler(),
                                           sigma_schedule=ref.sampler.KarrasSchedule(sigma_min=0.0001, sigma_max=3.0,
                                                                                     rho=9.0), clamp=False)
    B, N, steps = SAMP["B"], SAMP["N"], SAMP["steps"]
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(B, 1, 256, generator=g)
    emb = torch.randn(B, N, 768, generator=g)
    feats = torch.randn(B, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    out = {}
    for scale in (1.0, 1.5):
        kw = dict(embedding=emb, embedding_scale=scale, num_steps=steps)
        if cfg["model_params"]["multispeaker"]:
            kw["features"] = feats
        with torch.no_grad(), replay_randn_like({(B, 1, 256): list(step_noise)}):
            r = sampler(noise, **kw)
        out["samp_%s_scale%.1f" % (tag, scale)] = r.numpy()
    return out


def frontend_vectors():
    """LJSpeech text encoder / PL-BERT / duration encoder / durations / F0-N predictor through the reference modules,
    following Demo/Inference_LJSpeech.ipynb:280-311."""
    model, args, cfg = RH.build_reference_model("config.yml")
    ref = RH.load_reference()
    for i, k in enumerate(KEYS):
        synth.init_synthetic_(model[k], 10 + i)
    B, N, steps = E2E["B"], E2E["N"], E2E["steps"]
    g = torch.Generator().manual_seed(0)
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    lengths = torch.LongTensor([N] * B)
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    sampler = ref.sampler.DiffusionSampler(model.diffusion.diffusion, sampler=ref.sampler.ADPM2Sampler(),
                                           sigma_schedule=ref.sampler.KarrasSchedule(sigma_min=0.0001, sigma_max=3.0,
                                                                                     rho=9.0), clamp=False)
    with torch.no_grad():
        mask = torch.gt(torch.arange(N).unsqueeze(0) + 1, lengths.unsqueeze(1))
        t_en = model.text_encoder(tokens, lengths, mask)
        bert_dur = model.bert(tokens, attention_mask=(~mask).int())
        d_en = model.bert_encoder(bert_dur).transpose(-1, -2)
        with replay_randn_like({(B, 1, 256): list(step_noise)}):
            s_pred = sampler(noise, embedding=bert_dur, num_steps=steps, embedding_scale=1).squeeze(0)
        s = s_pred[:, 128:]
        d = model.predictor.text_encoder(d_en, s, lengths, mask)
        x, _ = model.predictor.lstm(d)
        duration = torch.sigmoid(model.predictor.duration_proj(x)).sum(axis=-1)
        pred_dur = torch.round(duration.squeeze()).clamp(min=1)
        pred_dur[-1] += 5
        aln = torch.zeros(N, int(pred_dur.sum()))
        c = 0
        for i in range(N):
            aln[i, c:c + int(pred_dur[i])] = 1
            c += int(pred_dur[i])
        en = d.transpose(-1, -2) @ aln.unsqueeze(0)
        F0, Nn = model.predictor.F0Ntrain(en, s)
    return {"fe_t_en": t_en.numpy(), "fe_bert_dur": bert_dur.numpy(), "fe_s_pred": s_pred.numpy(), "fe_d": d.numpy(),
            "fe_dur": pred_dur.numpy(), "fe_F0": F0.numpy(), "fe_N": Nn.numpy()}


def frontend_vectors_multispeaker_istftnet():
    """BASELINE.json configs[3] (LibriTTS multispeaker + iSTFTNet decoder) through the reference modules, following
    the `inference` cell of Demo/Inference_LibriTTS.ipynb: style mixing with the reference style, no +5 tail, and --
    because decoder.type != "hifigan" -- no one-frame shift of en / asr."""
    from oracle.make_golden import istftnet_decoder_override
    model, args, cfg = RH.build_reference_model("config_libritts.yml", overrides=istftnet_decoder_override(),
                                                replace_keys=("decoder",))
    assert cfg["model_params"]["multispeaker"] and cfg["model_params"]["decoder"]["type"] == "istftnet"
    ref = RH.load_reference()
    for i, k in enumerate(KEYS):
        synth.init_synthetic_(model[k], 10 + i)
    N, steps, alpha, beta = 7, 3, 0.3, 0.7
    g = torch.Generator().manual_seed(21)
    tokens = torch.randint(1, 178, (1, N), generator=g)
    tokens[:, 0] = 0
    lengths = torch.LongTensor([N])
    noise = torch.randn(1, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, 1, 1, 256, generator=g)
    ref_s = torch.randn(1, 256, generator=g)
    sampler = ref.sampler.DiffusionSampler(model.diffusion.diffusion, sampler=ref.sampler.ADPM2Sampler(),
                                           sigma_schedule=ref.sampler.KarrasSchedule(sigma_min=0.0001, sigma_max=3.0,
                                                                                     rho=9.0), clamp=False)
    with torch.no_grad():
        mask = torch.gt(torch.arange(N).unsqueeze(0) + 1, lengths.unsqueeze(1))
        t_en = model.text_encoder(tokens, lengths, mask)
        bert_dur = model.bert(tokens, attention_mask=(~mask).int())
        d_en = model.bert_encoder(bert_dur).transpose(-1, -2)
        with replay_randn_like({(1, 1, 256): list(step_noise)}):
            s_pred = sampler(noise, embedding=bert_dur, embedding_scale=1, features=ref_s, num_steps=steps).squeeze(1)
        s = beta * s_pred[:, 128:] + (1 - beta) * ref_s[:, 128:]
        rf = alpha * s_pred[:, :128] + (1 - alpha) * ref_s[:, :128]
        d = model.predictor.text_encoder(d_en, s, lengths, mask)
        x, _ = model.predictor.lstm(d)
        duration = torch.sigmoid(model.predictor.duration_proj(x)).sum(axis=-1)
        pred_dur = torch.round(duration.squeeze()).clamp(min=1)
        T = int(pred_dur.sum())
        aln = torch.zeros(N, T)
        c = 0
        for i in range(N):
            aln[i, c:c + int(pred_dur[i])] = 1
            c += int(pred_dur[i])
        en = d.transpose(-1, -2) @ aln.unsqueeze(0)
        F0, Nn = model.predictor.F0Ntrain(en, s)
        asr = t_en @ aln.unsqueeze(0)
        sine = torch.randn(1, 600 * T, 9, generator=g)
        taps = {}
        dec = model.decoder
        orig = dec.generator.stft.transform

        def tr(x):
            a, b = orig(x)
            taps["har"] = torch.cat([a, b], 1)
            return a, b
        dec.generator.stft.transform = tr
        with replay_randn_like({sine.shape: [sine]}):
            wave = dec(asr, F0, Nn, rf.squeeze().unsqueeze(0))
    # the 600*T-sample waveform is stored as per-frame RMS (T values) plus its first / last 2400 samples: enough to pin
    # the decoder call (same architecture as the LJSpeech decoder, pinned in full by dec_ljspeech_*) at fixture size
    w = wave.reshape(-1)
    return {"ms_s_pred": s_pred.numpy(), "ms_ref": rf.numpy(), "ms_dur": pred_dur.numpy(), "ms_F0": F0.numpy(),
            "ms_N": Nn.numpy(), "ms_asr_sum": asr.sum(dim=1).numpy(), "ms_wave_head": w[:2400].numpy(),
            "ms_wave_tail": w[-2400:].numpy(), "ms_wave_frame_rms": w.reshape(T, 600).pow(2).mean(dim=1).sqrt().numpy(),
            "ms_har_mag_mean": taps["har"][:, :11].mean(dim=2).numpy()}


def main():
    vec = {}
    vec.update(decoder_vectors("ljspeech", "config.yml"))
    vec.update(decoder_vectors("libritts", "config_libritts.yml"))
    vec.update(sampler_vectors("ljspeech", "config.yml"))
    vec.update(sampler_vectors("libritts", "config_libritts.yml"))
    vec.update(frontend_vectors())
    vec.update(frontend_vectors_multispeaker_istftnet())
    path = os.path.join(GOLDEN, "reference_vectors.npz")
    np.savez_compressed(path, **{k: np.asarray(v, dtype=np.float32) for k, v in vec.items()})
    print(path, os.path.getsize(path), sorted((k, v.shape) for k, v in vec.items()))
