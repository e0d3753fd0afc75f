"""TEST INFRASTRUCTURE ONLY -- fp64 numpy evaluation of the reference's mel front-end, independent of the product code.

The reference computes its log-mel with `torchaudio.transforms.MelSpectrogram(n_mels=80, n_fft=2048, win_length=1200,
hop_length=300)` and normalises `(log(1e-5 + mel) - (-4)) / 4` (meldataset.py:58-66; Demo/Inference_LibriTTS.ipynb
`preprocess`).  torchaudio is not installed in the build container, so what is restated here is torchaudio's DOCUMENTED
algorithm (torchaudio 2.x `transforms.Spectrogram` / `functional.spectrogram` / `functional.melscale_fbanks`), term by
term, in float64, with explicit loops over frames instead of torch.stft:

  Spectrogram defaults : power = 2, normalized = False, center = True, pad_mode = "reflect", onesided = True,
                         window = hann_window(win_length) (periodic), zero-padded on both sides to n_fft
                         (left = (n_fft - win_length) // 2) as torch.stft does for win_length < n_fft
  frames               : reflect-pad the signal by n_fft // 2 on both sides, frame t = padded[t * hop : t * hop + n_fft],
                         1 + L // hop frames
  MelScale defaults    : sample_rate = 16000 (the reference NEVER passes its 24 kHz rate: meldataset.py:58-59), f_min = 0,
                         f_max = sample_rate // 2, norm = None, mel_scale = "htk":
                           all_freqs = linspace(0, sample_rate // 2, n_freqs)
                           m_pts = linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2), hz_to_mel(f) = 2595 log10(1 + f / 700)
                           f_pts = 700 (10^(m / 2595) - 1); f_diff = f_pts[1:] - f_pts[:-1]
                           slopes[i, j] = f_pts[j] - all_freqs[i]
                           fb[i, m] = max(0, min(-slopes[i, m] / f_diff[m], slopes[i, m + 2] / f_diff[m + 1]))
                         mel = fb^T . power

This file is the checker the HIP front-end (styletts2_amd/style.py mel_spectrogram_engine) and the north_star's mel-L1
metric (tests/_util.py) are held to; its own outputs on seeded inputs are committed as tests/golden/mel_vectors.npz
(oracle/golden_mel.py) and it is cross-checked against a torch.stft evaluation in tests/test_style_cpu.py.  Parity with the
torchaudio BINARY remains unpinned (it cannot be imported here); the formulae above are what is pinned.
"""
import math

import numpy as np

MEL_MEAN, MEL_STD = -4.0, 4.0  # meldataset.py:60


def hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs=1025, n_mels=80, sample_rate=16000, f_min=0.0, f_max=None):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") in float64: [n_freqs, n_mels]."""
    f_max = float(sample_rate // 2) if f_max is None else float(f_max)
    all_freqs = np.linspace(0.0, float(sample_rate // 2), n_freqs)
    m_pts = np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    fb = np.zeros((n_freqs, n_mels))
    for m in range(n_mels):
        lo, ce, hi = f_pts[m], f_pts[m + 1], f_pts[m + 2]
        for i in range(n_freqs):
            down = (all_freqs[i] - lo) / (ce - lo)   # == -slopes[i, m] / f_diff[m]
            up = (hi - all_freqs[i]) / (hi - ce)     # == slopes[i, m + 2] / f_diff[m + 1]
            fb[i, m] = max(0.0, min(down, up))
    return fb


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(n) / n)


def power_spectrogram(wave, n_fft=2048, win_length=1200, hop_length=300):
    """wave [L] float -> |STFT|^2 [n_fft // 2 + 1, 1 + L // hop] in float64 (Spectrogram defaults, see above)."""
    x = np.asarray(wave, dtype=np.float64)
    L = x.shape[0]
    p = n_fft // 2
    assert L > p, "reflect padding needs more than n_fft / 2 samples"
    padded = np.concatenate([x[1:p + 1][::-1], x, x[-p - 1:-1][::-1]])
    win = np.zeros(n_fft)
    left = (n_fft - win_length) // 2
    win[left:left + win_length] = hann_periodic(win_length)
    n_frames = 1 + L // hop_length
    out = np.empty((n_fft // 2 + 1, n_frames))
    for t in range(n_frames):
        seg = padded[t * hop_length:t * hop_length + n_fft] * win
        spec = np.fft.rfft(seg)
        out[:, t] = spec.real ** 2 + spec.imag ** 2
    return out


def mel_spectrogram(wave, n_fft=2048, win_length=1200, hop_length=300, n_mels=80):
    """wave [L] or [B, L] (24 kHz) -> normalised log-mel [80, 1 + L // 300] / [B, 80, ...], float64."""
    w = np.asarray(wave, dtype=np.float64)
    if w.ndim == 2:
        return np.stack([mel_spectrogram(r, n_fft, win_length, hop_length, n_mels) for r in w])
    fb = mel_filterbank(n_fft // 2 + 1, n_mels)
    mel = fb.T @ power_spectrogram(w, n_fft, win_length, hop_length)
    return (np.log(1e-5 + mel) - MEL_MEAN) / MEL_STD


def mel_spectrogram_t(wave):
    """torch in, torch float32 out (tests' convenience): wave [..., L] -> [..., 80, frames]."""
    import torch
    w = wave.detach().cpu().double().numpy()
    lead = w.shape[:-1]
    m = mel_spectrogram(w.reshape(-1, w.shape[-1]))
    return torch.from_numpy(m.reshape(*lead, m.shape[-2], m.shape[-1])).float()
