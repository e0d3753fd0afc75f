"""Reference-audio style path on the GPU: the HIP kernels of csrc/st2_style.hip against their
contracts, StyleEncoder on the engine (C++ plan and per-kernel plan) against the golden vectors produced by the
unmodified reference module (oracle/golden_style.py -> tests/golden/style_vectors.npz) and the oracle, the mel front-end
against oracle/mel_ref.py's committed fixtures (tests/golden/mel_vectors.npz)."""
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, manifest
from oracle import golden_mel, mel_ref
from oracle import ops_ref as R
from oracle import st2_oracle as O
from styletts2_amd import _hooks, models, ops, style
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = {"small": dict(dim_in=16, style_dim=32, max_conv_dim=64, B=3, T=83, seed=21),
         "libritts": dict(dim_in=64, style_dim=128, max_conv_dim=512, B=2, T=120, seed=22)}


def test_style_kernels_match_contracts():
    g = torch.Generator().manual_seed(3)
    wave = torch.randn(2, 7013, generator=g)
    fr = ops.stft_frames(wave.to(DEV), 1200, 300, 600)
    assert torch.equal(fr.cpu(), R.stft_frames(wave, 1200, 300, 600))          # pure gather: bit-exact
    y = torch.randn(2, 2 * 37, 50, generator=g)
    p = ops.power_spectrum(y.to(DEV))
    assert torch.equal(p.cpu(), R.power_spectrum(y))                           # one rounding per op, contraction off
    x = torch.rand(3, 80, 41, generator=g) * 10
    got = ops.log_norm_(x.to(DEV).clone(), 1e-5, -4.0, 4.0)
    assert (got.cpu() - R.log_norm_(x.clone(), 1e-5, -4.0, 4.0)).abs().max().item() < 1e-6
    for (B, H, C, W) in [(2, 8, 5, 33), (1, 80, 16, 83), (2, 10, 64, 11)]:
        big = torch.randn(B, H + 2, C, W, generator=g)                         # strided interior view of a padded map
        xm = big[:, 1:H + 1]
        w, bias = torch.randn(C, 3, 3, generator=g), torch.randn(C, generator=g)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        ref = R.dwconv3x3s2(xm, w, bias, torch.empty(B, Ho, C, Wo))
        out = ops.dwconv3x3s2(big.to(DEV)[:, 1:H + 1], w.to(DEV), bias.to(DEV), torch.empty(B, Ho, C, Wo, device=DEV))
        assert (out.cpu() - ref).abs().max().item() < 1e-5
        refp = R.avgpool2x2(xm, torch.empty(B, H // 2, C, (W + 1) // 2))
        outp = ops.avgpool2x2(big.to(DEV)[:, 1:H + 1], torch.empty(B, H // 2, C, (W + 1) // 2, device=DEV))
        assert (outp.cpu() - refp).abs().max().item() < 1e-6


@pytest.mark.parametrize("plan", ["engine", "python"])
@pytest.mark.parametrize("tag", ["small", "libritts"])
def test_style_encoder_engine_matches_reference_vectors(tag, plan):
    with _hooks.override(plan=plan):
        _style_encoder_case(tag)


def _style_encoder_case(tag):
    c = CASES[tag]
    gold = np.load(os.path.join(GOLDEN, "style_vectors.npz"))["style_" + tag]
    enc = style.StyleEncoder(dim_in=c["dim_in"], style_dim=c["style_dim"], max_conv_dim=c["max_conv_dim"]).eval()
    synth.init_spectral_norm_(enc, c["seed"])
    g = torch.Generator().manual_seed(c["seed"])
    x = torch.randn(c["B"], 1, 80, c["T"], generator=g) * 0.8 - 0.2
    out = enc.to(DEV)(x.to(DEV))
    torch.cuda.synchronize()
    ops.check_status()
    out = out.cpu().numpy()
    assert out.shape == gold.shape
    err = np.abs(out - gold).max()
    assert err < 1e-5 * max(1.0, np.abs(gold).max()), err
    # a 10 s reference recording's map (801 frames): the xs conv path (rows >= 256 columns) against the oracle
    xl = torch.randn(1, 1, 80, 801, generator=g) * 0.8 - 0.2
    ref = O.style_encoder(enc.cpu().state_dict(), xl)
    got = enc.to(DEV)(xl.to(DEV)).cpu()
    assert (got - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


def test_mel_frontend_engine_matches_oracle_fixtures():
    """mel_spectrogram_engine against the committed outputs of oracle/mel_ref.py (fp64 evaluation of torchaudio's
    documented MelSpectrogram algorithm incl. the sample_rate = 16000 quirk) on the fixture generator's seeded waveforms:
    speech-like dynamics, full-scale input (|X|^2 ~ 1e5) and a length that is not a multiple of the hop."""
    gold = np.load(os.path.join(GOLDEN, "mel_vectors.npz"))
    for name, wave in golden_mel.waves().items():
        out = style.mel_spectrogram_engine(wave.to(DEV)).cpu().numpy()
        assert out.shape == gold[name].shape
        err = np.abs(out - gold[name]).max()
        assert err < 2e-4, (name, err)


def test_compute_style_engine_vs_oracle():
    man = manifest("libritts")
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    synth.init_spectral_norm_(model.style_encoder, 3)
    synth.init_spectral_norm_(model.predictor_encoder, 4)
    wave = torch.randn(2, 24000 * 3, generator=torch.Generator().manual_seed(0)) * 0.1
    ref = O.compute_style(model.style_encoder.state_dict(), model.predictor_encoder.state_dict(), wave)
    model.style_encoder.to(DEV)
    model.predictor_encoder.to(DEV)
    out = style.compute_style(model, wave.to(DEV))          # mel kernels + both encoders as C++ plans
    assert out.shape == (2, 256)
    assert (out.cpu() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    with _hooks.override(plan="python"):                    # per-kernel plan: bitwise the C++ plan
        assert torch.equal(style.compute_style(model, wave.to(DEV)), out)
