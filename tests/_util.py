"""Shared helpers for the parity tests."""
import json
import math
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Parity bars from BASELINE.json's north_star: "within 1e-3 mel L1 and 1e-4 waveform RMS".
WAVE_RMS_TOL = 1e-4
MEL_L1_TOL = 1e-3


def manifest(tag):
    from benchdata import manifest as _m  # benchdata/manifests/manifest_<tag>.json
    return _m(tag)


def decoder_kwargs(dc, hidden=512, style_dim=128, n_mels=80):
    kw = dict(dim_in=hidden, style_dim=style_dim, dim_out=n_mels,
              resblock_kernel_sizes=dc["resblock_kernel_sizes"], upsample_rates=dc["upsample_rates"],
              upsample_initial_channel=dc["upsample_initial_channel"],
              resblock_dilation_sizes=dc["resblock_dilation_sizes"],
              upsample_kernel_sizes=dc["upsample_kernel_sizes"], kind=dc["type"])
    if dc["type"] == "istftnet":
        kw.update(gen_istft_n_fft=dc["gen_istft_n_fft"], gen_istft_hop_size=dc["gen_istft_hop_size"])
    return kw


def rms(x):
    return x.detach().double().pow(2).mean().sqrt().item()


def phase_err_weighted(har_a, har_b, nb):
    """|wrap(phase_a - phase_b)| * |X|: the harmonic-STFT phase is ill-conditioned where |X| ~ 0 and wraps at
    +-pi, so it is compared on the unit circle weighted by the magnitude."""
    d = torch.remainder(har_a[:, nb:] - har_b[:, nb:] + math.pi, 2 * math.pi) - math.pi
    return (d.abs() * har_b[:, :nb]).max().item()


def mel_l1(wave_a, wave_b):
    """The second parity metric of BASELINE.json's north_star: L1 distance between the reference's normalised log-mel
    spectrograms ((log(1e-5 + mel) + 4) / 4 of MelSpectrogram(n_mels=80, n_fft=2048, win_length=1200, hop_length=300),
    meldataset.py:58-66) of two waveforms [..., L], evaluated on the CPU."""
    from oracle.mel_ref import mel_spectrogram_t  # fp64 evaluation of torchaudio's documented algorithm: not product code
    a = mel_spectrogram_t(wave_a.detach().cpu().float().reshape(-1, wave_a.shape[-1]))
    b = mel_spectrogram_t(wave_b.detach().cpu().float().reshape(-1, wave_b.shape[-1]))
    return (a - b).abs().mean().item()
