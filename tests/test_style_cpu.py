"""Reference-audio style path: the oracle's StyleEncoder against golden vectors produced by the
unmodified reference module (oracle/golden_style.py) and the engine's plan against both; the mel oracle
(oracle/mel_ref.py: fp64 evaluation of torchaudio's documented algorithm -- torchaudio itself is not installed here)
against its committed fixtures, a torch.stft evaluation and the front-end's defining properties; compute_style wiring and
the token table."""
import math
import os

import numpy as np
import pytest
import torch

from _plan_on_cpu import ops_on_cpu
from _util import GOLDEN, manifest
from oracle import golden_mel, mel_ref
from oracle import st2_oracle as O
from styletts2_amd import models, style, text_utils
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)

CASES = {"small": dict(dim_in=16, style_dim=32, max_conv_dim=64, B=3, T=83, seed=21),
         "libritts": dict(dim_in=64, style_dim=128, max_conv_dim=512, B=2, T=120, seed=22)}


@pytest.mark.parametrize("tag", ["small", "libritts"])
def test_style_encoder_matches_reference_vectors(tag):
    c = CASES[tag]
    gold = np.load(os.path.join(GOLDEN, "style_vectors.npz"))["style_" + tag]
    enc = style.StyleEncoder(dim_in=c["dim_in"], style_dim=c["style_dim"], max_conv_dim=c["max_conv_dim"]).eval()
    synth.init_spectral_norm_(enc, c["seed"])
    g = torch.Generator().manual_seed(c["seed"])
    x = torch.randn(c["B"], 1, 80, c["T"], generator=g) * 0.8 - 0.2
    out = O.style_encoder(enc.state_dict(), x).numpy()  # the oracle is pinned to the reference module's output ...
    assert out.shape == gold.shape
    assert np.abs(out - gold).max() < 2e-6 * max(1.0, np.abs(gold).max())
    # ... and so is the engine's launch plan (row-stacked Conv1d form of every Conv2d, (h, c, w) maps, pooled shortcut in the conv
    # epilogue) with the per-kernel CPU contracts substituted for the HIP wrappers
    with ops_on_cpu():
        plan = enc(x).numpy()
    assert plan.shape == gold.shape
    assert np.abs(plan - gold).max() < 1e-5 * max(1.0, np.abs(gold).max()), np.abs(plan - gold).max()


def test_style_encoder_engine_refuses_cpu_tensors():
    enc = style.StyleEncoder(dim_in=16, style_dim=32, max_conv_dim=64).eval()
    with pytest.raises(Exception, match="HIP device|no CPU path"):
        enc(torch.randn(1, 1, 80, 83))


def _mel_torch_stft(wave):
    """A second, independent evaluation of the same transform (torch.stft in fp64 with torchaudio's Spectrogram
    arguments), used only to cross-check the explicit-loop oracle's framing / padding / window conventions."""
    w = wave.double()
    spec = torch.stft(w, 2048, hop_length=300, win_length=1200, window=torch.hann_window(1200, periodic=True,
                      dtype=torch.float64), center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    fb = torch.from_numpy(mel_ref.mel_filterbank())
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return ((torch.log(1e-5 + mel) + 4.0) / 4.0).float()


def test_mel_oracle_reproduces_fixtures_and_agrees_with_torch_stft():
    gold = np.load(os.path.join(GOLDEN, "mel_vectors.npz"))
    for name, wave in golden_mel.waves().items():
        if name == "full_scale":
            continue  # same code path as tone_noise; kept for the GPU test, skipped here for CPU time
        m = mel_ref.mel_spectrogram_t(wave)
        assert np.abs(m.numpy() - gold[name]).max() < 1e-6, name          # the committed fixture is this oracle's output
        assert (m - _mel_torch_stft(wave)).abs().max().item() < 1e-5, name  # two independent evaluations agree


def test_mel_engine_plan_matches_oracle():
    """Frame gather + windowed DFT as a k=1 conv + power + filter bank + log on the CPU contracts vs the fp64 oracle."""
    g = torch.Generator().manual_seed(5)
    wave = torch.randn(2, 24000, generator=g) * 0.1 + 0.3 * torch.sin(torch.arange(24000) * 0.05)
    ref = mel_ref.mel_spectrogram_t(wave)
    with ops_on_cpu():
        out = style.mel_spectrogram_engine(wave)
    assert out.shape == ref.shape == (2, 80, 81)
    assert (out - ref).abs().max().item() < 2e-4, (out - ref).abs().max().item()


def test_mel_frontend_properties():
    sr, L = 24000, 24000
    t = torch.arange(L) / sr
    wave = 0.5 * torch.sin(2 * math.pi * 1000.0 * t)
    mel = mel_ref.mel_spectrogram_t(wave)
    assert mel.shape == (80, L // 300 + 1)
    raw = torch.exp(mel * style.MEL_STD + style.MEL_MEAN) - 1e-5
    fb = style.mel_filterbank()
    assert fb.shape == (1025, 80) and float(fb.min()) >= 0.0
    # the product's filter bank (a packed conv weight of the front-end, evaluated in fp32 like torchaudio evaluates its
    # own) against the oracle's fp64 one: fp32 rounding of the mel <-> Hz maps, nothing more
    assert (fb.double() - torch.from_numpy(mel_ref.mel_filterbank())).abs().max().item() < 1e-5
    # every FFT bin is covered by at most two overlapping triangles that sum to <= 1 (HTK scale, norm=None)
    assert float(fb.sum(dim=1).max()) <= 1.0 + 1e-5
    # the reference never passes sample_rate: the bank is laid out for 16 kHz, so a 1 kHz tone at 24 kHz (bin
    # 1000 / 24000 * 2048 = 85.3) lands in the filter whose 16 kHz-grid centre is 85.3 / 1024 * 8000 = 667 Hz
    peak = int(raw[:, 10:-10].mean(dim=1).argmax())
    mel_pts = torch.linspace(0, 2595.0 * math.log10(1 + 8000 / 700.0), 82)
    centres = (700.0 * (10.0 ** (mel_pts / 2595.0) - 1.0))[1:-1]
    assert abs(float(centres[peak]) - 667.0) < 40.0
    # linearity in power: doubling the amplitude adds log(4) / 4 to the normalised log-mel where the tone dominates
    mel2 = mel_ref.mel_spectrogram_t(2 * wave)
    assert abs(float((mel2 - mel)[peak, 20:60].mean()) - math.log(4.0) / 4.0) < 1e-3


def test_compute_style_and_builder_wiring():
    man = manifest("libritts")
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    assert isinstance(model.style_encoder, style.StyleEncoder) and isinstance(model.predictor_encoder, style.StyleEncoder)
    synth.init_spectral_norm_(model.style_encoder, 3)
    synth.init_spectral_norm_(model.predictor_encoder, 4)
    wave = torch.randn(2, 24000 * 2, generator=torch.Generator().manual_seed(0)) * 0.1
    with ops_on_cpu():
        ref_s = style.compute_style(model, wave)
        one = style.compute_style(model, wave[0])
    assert ref_s.shape == (2, 256) and bool(torch.isfinite(ref_s).all())
    assert torch.allclose(one, ref_s[:1], atol=1e-5)
    ref_t = O.compute_style(model.style_encoder.state_dict(), model.predictor_encoder.state_dict(), wave)
    assert torch.allclose(ref_s, ref_t, atol=2e-4), (ref_s - ref_t).abs().max()


def test_text_cleaner_table():
    tc = text_utils.TextCleaner()
    assert len(text_utils.SYMBOLS) == 178 and text_utils.SYMBOL_TO_ID["$"] == 0
    ids = tc("ðɪs ɪz ɐ tˈɛst!")
    assert len(ids) == 15 and all(0 < i < 178 for i in ids)
    assert tc.encode("a") == [0, text_utils.SYMBOL_TO_ID["a"]]
    assert tc("a#b") == [text_utils.SYMBOL_TO_ID["a"], text_utils.SYMBOL_TO_ID["b"]]  # unknown symbols are dropped
