"""Reference-audio style path: mel front-end + the two StyleEncoders that turn a reference
recording into `ref_s` [B, 256] for the multi-speaker models (Demo/Inference_LibriTTS.ipynb:100-111 `compute_style`).

    ref_s = compute_style(model, wave_24k)          # cat(style_encoder(mel), predictor_encoder(mel))

`StyleEncoder` keeps the reference's state_dict layout key for key (models.py:139-164 on top of ResBlk :97-137 and
LearnedDownSample :27-42, all under old-style `torch.nn.utils.spectral_norm`: `weight_orig` / `weight_u` /
`weight_v`), so `load_checkpoint` fills it from the published checkpoints.  In eval mode spectral norm is a fixed
rescale (no power iteration), folded once per load exactly as the reference computes it: sigma = u . (W_mat v),
W = weight_orig / sigma.

Both the encoders and the mel front-end run on the engine's HIP kernels (`forward` = one `st2_style_forward` call into the
C++ launch plan, csrc/st2_engine.hip style_plan; `mel_spectrogram_engine` = five kernel-level calls):
  * feature maps are stored (h, c, w) with one zero row above and below, so three consecutive rows ARE the 3C-channel
    input of a Conv1d over the width: every 3x3 Conv2d is one split-f16 MFMA `st2_conv1d` per utterance (batch = image
    rows, LeakyReLU as the conv prologue, residual add and 1/sqrt(2) in the epilogue), the 5x5 valid conv the same with
    5 stacked rows, the 1x1 shortcut a k=1 conv; the depthwise stride-2 conv and the 2x2 average pool are
    `st2_dwconv3x3s2` / `st2_avgpool2x2`; AdaptiveAvgPool + LeakyReLU + Linear = `st2_mean_tokens` + a k=1 conv with a
    LeakyReLU prologue;
  * mel: `st2_stft_frames` (reflect-padded frame columns) -> windowed DFT as an exact-fp32 MFMA k=1 conv
    [2050][1200] -> `st2_power_spectrum` -> mel filter bank as a k=1 conv [80][1025] -> `st2_log_norm`.
There is no PyTorch forward behind any of this: the nn.Modules below are parameter holders with the reference's keys, a
CPU tensor raises in the kernel wrappers.  The per-kernel Python plan (`StyleEncoder._forward_kernels`) is the tests' tap
path (`_hooks.override(plan="python")`) and what the CPU plan tests step through; the C++ plan is bitwise equal to it on
the GPU (profiles/archive/r03/r03a_style_plan.log).

The mel front-end implements `torchaudio.transforms.MelSpectrogram(n_mels=80, n_fft=2048, win_length=1200,
hop_length=300)` with torchaudio's defaults (power 2, periodic Hann window zero-padded to n_fft, centre + reflect
padding, HTK mel scale, no filter normalisation) INCLUDING the quirk that the reference never passes `sample_rate`
(meldataset.py:58-59): torchaudio's default 16 000 applies, i.e. the filter bank spans 0-8 kHz of the bin grid.
torchaudio is not installed in the build container; the front-end is pinned to oracle/mel_ref.py, an independent fp64
numpy evaluation of torchaudio's documented formulae (fixtures tests/golden/mel_vectors.npz), and the encoders to the
reference modules themselves (tests/golden/style_vectors.npz).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import transient_state
from . import _hooks, ops
from . import weights as W

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

    def folded_host(self):
        """Eval-mode spectral norm: weight_orig / (u . (W_mat v)); no power iteration (spectral_norm.py compute_weight
        with do_power_iteration=False) -- on the host in fp32 (what the packed engine weights are built from: identical
        bits whatever device the module lives on)."""
        w = self.weight_orig.detach().float().cpu()
        sigma = torch.dot(self.weight_u.float().cpu(), torch.mv(w.reshape(w.shape[0], -1), self.weight_v.float().cpu()))
        return w / sigma


class _LearnedDownSample(nn.Module):
    """models.py:27-42, layer_type 'half': depthwise 3x3, stride 2."""

    def __init__(self, dim_in):
        super().__init__()
        self.conv = _SNConv2d(dim_in, dim_in, 3, stride=2, padding=1, groups=dim_in)


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



@transient_state
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

        self._pk = None

    # -- packed-weight cache (invalidated by .to() / load_state_dict, also when a parent module recurses here) ----------
    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._pk = None
        return super().load_state_dict(W.strip_module_prefix(state_dict), *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._pk = None
        return super()._load_from_state_dict(*a, **k)

    def refresh(self):
        self._pk = None

    @staticmethod
    def _rows_as_channels(w2d):
        """Conv2d weight [Co][Ci][kh][kw] -> Conv1d weight [Co][kh*Ci][kw] acting on kh stacked image rows (channel
        index dh*Ci + ci), see the module docstring."""
        co, ci, kh, kw = w2d.shape
        return w2d.permute(0, 2, 1, 3).reshape(co, kh * ci, kw).contiguous()

    def _prepare(self, device):
        d = lambda t: None if t is None else t.detach().float().contiguous().to(device)
        pk = type("PackedStyleEncoder", (), {})()
        pk.device = device
        first = self.shared[0]
        pk.w0, pk.b0 = d(self._rows_as_channels(first.folded_host())), d(first.bias)            # [C0][3][3], direct conv
        pk.blocks = []
        for blk in list(self.shared)[1:5]:
            b = type("PackedResBlk", (), {})()
            b.c_in, b.c_out = blk.conv1.weight_orig.shape[1], blk.conv2.weight_orig.shape[0]
            b.w1 = W.pack_conv_auto(self._rows_as_channels(blk.conv1.folded_host())).to(device)
            b.b1 = d(blk.conv1.bias)
            b.w2 = W.pack_conv_auto(self._rows_as_channels(blk.conv2.folded_host())).to(device)
            b.b2 = d(blk.conv2.bias)
            b.wd = d(blk.downsample_res.conv.folded_host().reshape(b.c_in, 3, 3))
            b.bd = d(blk.downsample_res.conv.bias)
            b.wsc = W.pack_conv_auto(blk.conv1x1.folded_host().reshape(b.c_out, b.c_in, 1)).to(device) \
                if blk.learned_sc else None
            pk.blocks.append(b)
        last = self.shared[6]
        pk.w5 = W.pack_conv_auto(self._rows_as_channels(last.folded_host())).to(device)          # [C][5*C][5]
        pk.b5 = d(last.bias)
        pk.c_last = last.weight_orig.shape[0]
        pk.wl = W.pack_linear_auto(self.unshared.weight.detach().float().cpu()).to(device)
        pk.bl = d(self.unshared.bias)
        self._pk = pk
        return pk

    @torch.no_grad()
    def forward(self, x):
        """mel [B, 1, 80, T] -> style [B, style_dim]: one `st2_style_forward` call into the C++ launch plan."""
        if x.device.type == "cuda" and _hooks.plan == "engine":
            from . import engine
            from .text import _cached_engine
            eng = _cached_engine(self, "_engine", [self], lambda: engine.build_style_engine(self, None, x.device))
            return eng.style_forward(0, x)
        return self._forward_kernels(x)

    @torch.no_grad()
    def _forward_kernels(self, x):
        """The same plan kernel by kernel from Python (tests' tap path; module docstring)."""
        x = x.float()
        dev = x.device
        pk = self._pk if (self._pk is not None and self._pk.device == dev) else self._prepare(dev)
        B, one, H, Wd = x.shape
        assert one == 1 and H % 16 == 0 and Wd >= 80, "StyleEncoder needs [B, 1, 16k, T >= 80], got %s" % (tuple(x.shape),)
        new_map = lambda h, c, w: torch.zeros((B, h + 2, c, w), device=dev, dtype=torch.float32)  # zero rows 0 and h+1

        def rows(P, b, k):
            """Rows h-1 .. h+k-2 of utterance b's padded map stacked along the channels: [H][k*C][W] (overlapping view)."""
            _, hp, c, w = P.shape
            return torch.as_strided(P[b], (hp - k + 1, k * c, w), (c * w, w, 1))

        m0 = new_map(H, 1, Wd)
        ops.copy_ncl(x.reshape(B, H, Wd).contiguous(), m0[:, 1:H + 1, 0])
        C0 = pk.w0.shape[0]
        P = new_map(H, C0, Wd)
        for b in range(B):
            ops.conv1d_direct(rows(m0, b, 3), pk.w0, pk.b0, 1, 1, out=P[b, 1:H + 1])
        for blk in pk.blocks:
            C, Co = blk.c_in, blk.c_out
            Ho, Wo = H // 2, (Wd + 1) // 2
            # shortcut: 1x1 conv at full resolution, then the 2x2 average (models.py:118-123)
            if blk.wsc is not None:
                S = torch.empty((B, H, Co, Wd), device=dev, dtype=torch.float32)
                for b in range(B):
                    ops.conv1d(P[b, 1:H + 1], blk.wsc, Co, 1, out=S[b])
            else:
                S = P[:, 1:H + 1]
            SC = torch.empty((B, Ho, Co, Wo), device=dev, dtype=torch.float32)
            ops.avgpool2x2(S, SC)
            # residual: leaky -> conv1 3x3 -> depthwise stride-2 3x3 -> leaky -> conv2 3x3 (models.py:125-135)
            R1 = torch.empty((B, H, C, Wd), device=dev, dtype=torch.float32)
            for b in range(B):
                ops.conv1d(rows(P, b, 3), blk.w1, C, 3, pad_left=1, bias=blk.b1, pro=ops.PRO_LEAKY, slope=0.2, out=R1[b])
            P2 = new_map(Ho, C, Wo)
            ops.dwconv3x3s2(R1, blk.wd, blk.bd, P2[:, 1:Ho + 1])
            Pn = new_map(Ho, Co, Wo)
            for b in range(B):  # (shortcut + residual) / sqrt(2) in the epilogue
                ops.conv1d(rows(P2, b, 3), blk.w2, Co, 3, pad_left=1, bias=blk.b2, pro=ops.PRO_LEAKY, slope=0.2,
                           res=SC[b], div=math.sqrt(2), out=Pn[b, 1:Ho + 1])
            P, H, Wd = Pn, Ho, Wo
        # LeakyReLU -> 5x5 valid conv -> global average -> LeakyReLU -> Linear (models.py:151-163)
        Cl = pk.c_last
        assert H == 5, "the 5x5 valid conv expects a 5-row map (80 mel bins), got %d" % H
        Fm = torch.empty((B, Cl, Wd - 4), device=dev, dtype=torch.float32)
        for b in range(B):
            ops.conv1d(P[b, 1:6].reshape(1, 5 * P.shape[2], Wd), pk.w5, Cl, 5, pad_left=0, L_out=Wd - 4, bias=pk.b5,
                       pro=ops.PRO_LEAKY, slope=0.2, out=Fm[b:b + 1])
        m = ops.mean_tokens(Fm)                                                               # [B, Cl]
        s = ops.conv1d(m.reshape(B, Cl, 1), pk.wl, self.unshared.out_features, 1, bias=pk.bl, pro=ops.PRO_LEAKY, slope=0.2)
        return s.reshape(B, -1)


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


_MEL_PACK = {}


def _mel_pack(device, n_fft, win_length, n_mels):
    """Exact-fp32 MFMA conv weights of the front-end, built once per device: the windowed DFT restricted to the
    win_length taps where the zero-padded periodic Hann window is non-zero (rows k <= n_fft/2: w[n] cos(2 pi k n / n_fft),
    rows n_fft/2 + 1 + k: -w[n] sin(...), n counted in the padded frame) and the mel filter bank, both k=1 convs."""
    key = (str(device), n_fft, win_length, n_mels)
    pk = _MEL_PACK.get(key)
    if pk is None:
        left = (n_fft - win_length) // 2
        win = torch.hann_window(win_length, periodic=True, dtype=torch.float64)
        n = torch.arange(win_length, dtype=torch.float64) + left
        k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
        ang = 2.0 * math.pi * torch.outer(k, n) / n_fft
        dft = torch.cat([torch.cos(ang) * win, -torch.sin(ang) * win]).float()               # [2K, win]
        fb = mel_filterbank(n_fft // 2 + 1, n_mels).t().contiguous()                         # [n_mels, K]
        pk = (W.pack_conv(dft.unsqueeze(-1)).to(device), W.pack_conv(fb.unsqueeze(-1)).to(device))
        _MEL_PACK[key] = pk
    return pk


@torch.no_grad()
def mel_spectrogram_engine(wave, n_fft=2048, win_length=1200, hop_length=300, n_mels=80):
    """wave [B, L] (24 kHz) -> normalised log-mel [B, 80, 1 + L // 300]: (log(1e-5 + mel) + 4) / 4 (meldataset.py:58-66;
    Demo/Inference_LibriTTS.ipynb `preprocess`) on the HIP kernels (module docstring)."""
    wave = wave.float().contiguous()
    dft_w, fb_w = _mel_pack(wave.device, n_fft, win_length, n_mels)
    K = n_fft // 2 + 1
    frames = ops.stft_frames(wave, win_length, hop_length, n_fft // 2 - (n_fft - win_length) // 2)
    spec = ops.conv1d(frames, dft_w, 2 * K, 1)
    mel = ops.conv1d(ops.power_spectrum(spec), fb_w, n_mels, 1)
    return ops.log_norm_(mel, 1e-5, MEL_MEAN, MEL_STD)


def _style_engine(model, dev):
    """The st2_engine handle holding BOTH style encoders, packed once per (weights, device)."""
    from . import engine
    mods = [model.style_encoder, model.predictor_encoder]
    stamp = tuple((p.data_ptr(), p._version) for m in mods for p in list(m.parameters()) + list(m.buffers()))
    cached = getattr(model.style_encoder, "_plan_engine", None)
    if cached is None or cached[0] != stamp or not engine.same_device(cached[1], dev):
        engine.replaced(cached[1] if cached else None, "style")
        cached = (stamp, engine.build_style_engine(model.style_encoder, model.predictor_encoder, dev))
        model.style_encoder._plan_engine = cached
    return cached[1]


@torch.no_grad()
def compute_style(model, wave):
    """`compute_style` of Demo/Inference_LibriTTS.ipynb:100-111 minus the file I/O: wave [L] or [B, L] at 24 kHz
    (already trimmed; the notebook trims with librosa.effects.trim(top_db=30) on the host) -> ref_s [B, 256]."""
    if wave.dim() == 1:
        wave = wave.unsqueeze(0)
    mel = mel_spectrogram_engine(wave)
    if wave.device.type == "cuda" and _hooks.plan == "engine":  # both encoders as C++ launch plans (st2_style_forward)
        eng = _style_engine(model, wave.device)
        return torch.cat([eng.style_forward(0, mel), eng.style_forward(1, mel)], dim=1)
    mel = mel.unsqueeze(1)                                         # [B, 1, 80, T]
    return torch.cat([model.style_encoder(mel), model.predictor_encoder(mel)], dim=1)
