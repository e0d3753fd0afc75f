"""Reference-audio style path: mel front-end + the two StyleEncoders that turn a reference
recording into `ref_s` [B, 256] for the multi-speaker models (Demo/Inference_LibriTTS.ipynb:100-111 `compute_style`).

    ref_s = compute_style(model, wave_24k)          # cat(style_encoder(mel), predictor_encoder(mel))

`StyleEncoder` keeps the reference's state_dict layout key for key (models.py:139-164 on top of ResBlk :97-137 and
LearnedDownSample :27-42, all under old-style `torch.nn.utils.spectral_norm`: `weight_orig` / `weight_u` /
`weight_v`), so `load_checkpoint` fills it from the published checkpoints.  It runs once per speaker, outside the
timed text->waveform path, on PyTorch-ROCm ops (F.conv2d / avg_pool2d): plumbing, not a hand-written kernel --
SURVEY.md marks the row "next".  In eval mode spectral norm is a fixed rescale (no power iteration), folded once per
load exactly as the reference computes it: sigma = u . (W_mat v), W = weight_orig / sigma.

The mel front-end restates `torchaudio.transforms.MelSpectrogram(n_mels=80, n_fft=2048, win_length=1200,
hop_length=300)` with torchaudio's defaults (power 2, periodic Hann window zero-padded to n_fft, centre + reflect
padding, HTK mel scale, no filter normalisation) INCLUDING the quirk that the reference never passes `sample_rate`
(meldataset.py:58-59): torchaudio's default 16 000 applies, i.e. the filter bank spans 0-8 kHz of the bin grid.
torchaudio is not installed in the build container, so this front-end is "parity unpinned" (no golden vector);
the encoders themselves are pinned against the reference modules (tests/golden/style_vectors.npz).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

MEL_MEAN, MEL_STD = -4.0, 4.0  # meldataset.py:60


class _SNConv2d(nn.Module):
    """Parameter holder with the state_dict keys of `spectral_norm(nn.Conv2d(...))` (torch.nn.utils.spectral_norm:
    `weight_orig`, `weight_u`, `weight_v`, `bias`)."""

    def __init__(self, c_in, c_out, ks, stride=1, padding=0, groups=1, bias=True):
        super().__init__()
        kh, kw = (ks, ks) if isinstance(ks, int) else ks
        self.stride, self.padding, self.groups = stride, padding, groups
        self.weight_orig = nn.Parameter(torch.randn(c_out, c_in // groups, kh, kw) * 0.05)
        self.register_buffer("weight_u", F.normalize(torch.randn(c_out), dim=0))
        self.register_buffer("weight_v", F.normalize(torch.randn(c_in // groups * kh * kw), dim=0))
        self.bias = nn.Parameter(torch.zeros(c_out)) if bias else None

    def folded(self):
        """Eval-mode spectral norm: weight_orig / (u . (W_mat v)); no power iteration (spectral_norm.py compute_weight
        with do_power_iteration=False)."""
        w = self.weight_orig
        sigma = torch.dot(self.weight_u, torch.mv(w.reshape(w.shape[0], -1), self.weight_v))
        return w / sigma

    def forward(self, x):
        return F.conv2d(x, self.folded(), self.bias, self.stride, self.padding, 1, self.groups)


class _LearnedDownSample(nn.Module):
    """models.py:27-42, layer_type 'half': depthwise 3x3, stride 2."""

    def __init__(self, dim_in):
        super().__init__()
        self.conv = _SNConv2d(dim_in, dim_in, 3, stride=2, padding=1, groups=dim_in)

    def forward(self, x):
        return self.conv(x)


class _ResBlk(nn.Module):
    """models.py:97-137 with normalize=False, downsample='half'."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.downsample_res = _LearnedDownSample(dim_in)
        self.learned_sc = dim_in != dim_out
        self.conv1 = _SNConv2d(dim_in, dim_in, 3, 1, 1)
        self.conv2 = _SNConv2d(dim_in, dim_out, 3, 1, 1)
        if self.learned_sc:
            self.conv1x1 = _SNConv2d(dim_in, dim_out, 1, 1, 0, bias=False)

    @staticmethod
    def _avg_half(x):  # DownSample('half'), models.py:72-75: replicate the last column when the width is odd
        if x.shape[-1] % 2 != 0:
            x = torch.cat([x, x[..., -1].unsqueeze(-1)], dim=-1)
        return F.avg_pool2d(x, 2)

    def forward(self, x):
        sc = self.conv1x1(x) if self.learned_sc else x
        sc = self._avg_half(sc)
        r = self.conv1(F.leaky_relu(x, 0.2))
        r = self.downsample_res(r)
        r = self.conv2(F.leaky_relu(r, 0.2))
        return (sc + r) / math.sqrt(2)


class StyleEncoder(nn.Module):
    """models.py:139-164: mel [B, 1, 80, T] -> style [B, style_dim].  T >= 80 frames (the 5x5 valid conv after four
    halvings needs a 5-wide map)."""

    def __init__(self, dim_in=48, style_dim=48, max_conv_dim=384):
        super().__init__()
        blocks = [_SNConv2d(1, dim_in, 3, 1, 1)]
        dim_out = dim_in
        for _ in range(4):
            dim_out = min(dim_in * 2, max_conv_dim)
            blocks.append(_ResBlk(dim_in, dim_out))
            dim_in = dim_out
        blocks += [nn.LeakyReLU(0.2), _SNConv2d(dim_out, dim_out, 5, 1, 0), nn.AdaptiveAvgPool2d(1), nn.LeakyReLU(0.2)]
        self.shared = nn.Sequential(*blocks)
        self.unshared = nn.Linear(dim_out, style_dim)

    @torch.no_grad()
    def forward(self, x):
        h = self.shared(x.float())
        return self.unshared(h.view(h.size(0), -1))


# ---- mel front-end -------------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs=1025, n_mels=80, sample_rate=16000, f_min=0.0, f_max=None):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated: [n_freqs, n_mels] triangles."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # [n_freqs, n_mels + 2]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def mel_spectrogram(wave, n_fft=2048, win_length=1200, hop_length=300, n_mels=80):
    """wave [..., L] (24 kHz) -> normalised log-mel [..., 80, 1 + L // 300]: (log(1e-5 + mel) + 4) / 4
    (meldataset.py:58-66; Demo/Inference_LibriTTS.ipynb `preprocess`)."""
    wave = wave.float()
    window = torch.hann_window(win_length, periodic=True, device=wave.device)
    spec = torch.stft(wave, n_fft, hop_length=hop_length, win_length=win_length, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                        # [..., 1025, frames]
    fb = mel_filterbank(n_fft // 2 + 1, n_mels).to(wave.device)
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    return (torch.log(1e-5 + mel) - MEL_MEAN) / MEL_STD


@torch.no_grad()
def compute_style(model, wave):
    """`compute_style` of Demo/Inference_LibriTTS.ipynb:100-111 minus the file I/O: wave [L] or [B, L] at 24 kHz
    (already trimmed; the notebook trims with librosa.effects.trim(top_db=30) on the host) -> ref_s [B, 256]."""
    if wave.dim() == 1:
        wave = wave.unsqueeze(0)
    mel = mel_spectrogram(wave).unsqueeze(1)                       # [B, 1, 80, T]
    return torch.cat([model.style_encoder(mel), model.predictor_encoder(mel)], dim=1)
