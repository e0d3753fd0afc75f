"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

Per-kernel fp32 restatements, in plain PyTorch-CPU ops, of the contracts declared in
include/st2.h.  Same call signatures as styletts2_amd.ops so that tests can (a) compare each
HIP kernel against its contract on the GPU box and (b) run the host-side layer plans on CPU by
substituting these functions for the HIP wrappers (plan / weight-packing check without a GPU).
"""
import math

import torch
import torch.nn.functional as F

PRO_NONE, PRO_LEAKY, PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE, PRO_SNAKE, PRO_COLNORM = range(6)
ACT_NONE, ACT_GELU, ACT_EXP_SIN, ACT_TANH, ACT_LEAKY, ACT_GELU_TANH = range(6)


def _snake(u, alpha):
    a = alpha.view(1, -1, 1)
    return u + (1.0 / a) * torch.sin(a * u) ** 2


def conv1d(x, wt, C_out, ks, *, dil=1, pad_left=0, L_out=None, bias=None, out=None,
           pro=PRO_NONE, slope=0.0, stats=None, gamma=None, beta=None, gamma_plus_one=False, alpha=None,
           res=None, res_shift=0, res2=None, div=1.0, act=ACT_NONE, act_split=0, act_slope=0.0, want_stats=False,
           gb_seg=0):
    y = _conv1d(x, wt, C_out, ks, dil=dil, pad_left=pad_left, L_out=L_out, bias=bias, out=out, pro=pro, slope=slope,
                stats=stats, gamma=gamma, beta=beta, gamma_plus_one=gamma_plus_one, alpha=alpha, res=res,
                res_shift=res_shift, res2=res2, div=div, act=act, act_split=act_split, act_slope=act_slope,
                gb_seg=gb_seg)
    return (y, instnorm_stats(y)) if want_stats else y


def activate(x, *, pro=PRO_NONE, slope=0.0, stats=None, gamma=None, beta=None, gamma_plus_one=False, alpha=None, gb_seg=0):
    """pro(x): the prologue of the conv contract = what st2_act_split materialises (before the x8 scale + hi/lo split).
    gb_seg > 0 (PRO_COLNORM, x [1, C, L]): gamma / beta [G, C], row l // gb_seg applies at column l."""
    u = x
    if pro == PRO_LEAKY:
        u = F.leaky_relu(x, slope)
    elif pro in (PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE):
        n = (x - stats[:, :, 0:1]) * stats[:, :, 1:2]
        u = (1.0 + gamma.unsqueeze(-1)) * n + beta.unsqueeze(-1)
        u = F.leaky_relu(u, slope) if pro == PRO_ADAIN_LEAKY else _snake(u, alpha)
    elif pro == PRO_SNAKE:
        u = _snake(x, alpha)
    elif pro == PRO_COLNORM:
        n = (x - stats[:, :, 0].unsqueeze(1)) * stats[:, :, 1].unsqueeze(1)
        if gb_seg:
            rows = torch.arange(x.shape[-1]) // gb_seg
            g, bt = gamma[rows].t().unsqueeze(0), beta[rows].t().unsqueeze(0)   # [1, C, L]
        else:
            g, bt = gamma.unsqueeze(-1), beta.unsqueeze(-1)
        if gamma_plus_one:
            g = 1.0 + g
        u = n * g + bt
    return u


def _conv1d(x, wt, C_out, ks, *, dil=1, pad_left=0, L_out=None, bias=None, out=None,
            pro=PRO_NONE, slope=0.0, stats=None, gamma=None, beta=None, gamma_plus_one=False, alpha=None,
            res=None, res_shift=0, res2=None, div=1.0, act=ACT_NONE, act_split=0, act_slope=0.0, gb_seg=0):
    B, C_in, L_in = x.shape
    if L_out is None:
        L_out = L_in
    u = activate(x, pro=pro, slope=slope, stats=stats, gamma=gamma, beta=beta, gamma_plus_one=gamma_plus_one,
                 alpha=alpha, gb_seg=gb_seg)
    if hasattr(wt, "wq"):  # split-f16 packing (st2_conv1d_f16s): operands are hi + lo of v * scale; lo*lo dropped
        w = wt.dense()
        xs = 8.0 if pro in (PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE, PRO_COLNORM) else 1.0  # ops.x_scale_for(pro)
        hi = (u * xs).half().float()
        u = (hi + ((u * xs) - hi).half().float()) / xs
    else:
        w = wt[:, :C_out].reshape(C_in, ks, C_out).permute(2, 0, 1).contiguous()
    need = (L_out - 1) + (ks - 1) * dil + 1  # input span [ -pad_left, need - pad_left )
    pad_right = max(0, need - pad_left - L_in)
    up = F.pad(u, (pad_left, pad_right))
    y = F.conv1d(up, w, None, dilation=dil)[:, :, :L_out]
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    if res is not None:
        r = res
        if res_shift:
            r = res.repeat_interleave(1 << res_shift, dim=2)[:, :, :L_out]
        y = y + r
    if res2 is not None:
        y = res2 + y
    if div != 1.0:
        y = y / div
    if act == ACT_GELU:
        y = F.gelu(y)
    elif act == ACT_EXP_SIN:
        y = torch.cat([torch.exp(y[:, :act_split]), torch.sin(y[:, act_split:])], dim=1)
    elif act == ACT_TANH:
        y = torch.tanh(y)
    elif act == ACT_LEAKY:
        y = F.leaky_relu(y, act_slope)
    elif act == ACT_GELU_TANH:
        y = F.gelu(y, approximate="tanh")
    if out is not None:
        out.copy_(y)
        return out
    return y


def conv1d_direct(x, w, bias, stride, pad, L_out=None, out=None):
    y = F.conv1d(x, w, bias, stride=stride, padding=pad)
    if L_out is not None:
        y = y[:, :, :L_out]
    if out is not None:
        out.copy_(y)
        return out
    return y


def phase_split(x, stride, pad, Lu):
    B, Cc, L = x.shape
    need = Lu * stride
    xpad = F.pad(x, (pad, max(0, need - pad - L)))[:, :, :need]
    return xpad.reshape(B, Cc, Lu, stride).permute(0, 1, 3, 2).reshape(B, Cc * stride, Lu).contiguous()


def instnorm_stats(x, eps=1e-5, out=None):
    xd = x.double()
    mean = xd.mean(dim=2)
    var = xd.var(dim=2, unbiased=False)
    st = torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=-1).float()
    if out is not None:
        out.copy_(st)
        return out
    return st


def colnorm_stats(x, eps=1e-5, out=None):
    xd = x.double()
    mean = xd.mean(dim=1)
    var = xd.var(dim=1, unbiased=False)
    st = torch.stack([mean, 1.0 / torch.sqrt(var + eps)], dim=-1).float()
    if out is not None:
        out.copy_(st)
        return out
    return st


def style_fc(s, wt, bias, act=ACT_NONE, out=None):
    h = s @ wt
    if bias is not None:
        h = h + bias
    if act == ACT_GELU:
        h = F.gelu(h)
    if out is not None:
        out.copy_(h)
        return out
    return h


def convt_interleave(phases, C_out, stride, pad, L_raw, bias=None, add=None, reflect_left=False, out=None,
                     want_stats=False):
    y = _convt_interleave(phases, C_out, stride, pad, L_raw, bias=bias, add=add, reflect_left=reflect_left, out=out)
    return (y, instnorm_stats(y)) if want_stats else y


def _convt_interleave(phases, C_out, stride, pad, L_raw, bias=None, add=None, reflect_left=False, out=None):
    B, RC, Lq = phases.shape
    ph = phases.reshape(B, stride, C_out, Lq)
    full = ph.permute(0, 2, 3, 1).reshape(B, C_out, Lq * stride)  # index q*stride + r
    y = full[:, :, pad:pad + L_raw]
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    if reflect_left:
        y = torch.cat([y[:, :, 1:2], y], dim=2)
    if add is not None:
        y = y + add
    if out is not None:
        out.copy_(y)
        return out
    return y


def adain_leaky_pool(x, stats, gamma, beta, slope, w, bias, out=None):
    n = (x - stats[:, :, 0:1]) * stats[:, :, 1:2]
    u = F.leaky_relu((1.0 + gamma.unsqueeze(-1)) * n + beta.unsqueeze(-1), slope)
    Cc = x.shape[1]
    y = F.conv_transpose1d(u, w.view(Cc, 1, 3), bias, stride=2, padding=1, output_padding=1, groups=Cc)
    if out is not None:
        out.copy_(y)
        return out
    return y


def har_source(f0, U, noise, lin_w, lin_b, sine_amp=0.1, noise_std=0.003, voiced_threshold=10.0,
               sample_rate=24000.0):
    """Restates SineGen/SourceModuleHnNSF (Modules/istftnet.py:141-247,283-297) with torch ops in the
    reference's own order (the frame-rate shortcut of SURVEY.md App. A.1 step 4 is NOT used here)."""
    B, Fr = f0.shape
    H = noise.shape[2]
    f0u = f0[:, None].repeat_interleave(U, dim=2).transpose(1, 2)  # nearest x U  [B, L, 1]
    fn = f0u * torch.arange(1, H + 1, dtype=torch.float32).view(1, 1, H)
    rad = (fn / sample_rate) % 1
    rad = F.interpolate(rad.transpose(1, 2), scale_factor=1 / U, mode="linear").transpose(1, 2)
    phase = torch.cumsum(rad, dim=1) * 2 * math.pi
    phase = F.interpolate(phase.transpose(1, 2) * U, scale_factor=U, mode="linear").transpose(1, 2)
    sines = torch.sin(phase) * sine_amp
    uv = (f0u > voiced_threshold).float()
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    sw = sines * uv + noise_amp * noise
    return torch.tanh(sw @ lin_w.view(H, 1) + lin_b.view(1)).squeeze(-1)


def stft_mag_phase(x, n_fft, hop):
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
    X = torch.stft(x, n_fft, hop, n_fft, window=win, return_complex=True)
    return torch.cat([X.abs(), X.angle()], dim=1)


def istft(sp, n_fft, hop):
    nb = n_fft // 2 + 1
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
    y = torch.istft(sp[:, :nb] * torch.exp(sp[:, nb:] * 1j), n_fft, hop, n_fft, window=win)
    return y.unsqueeze(-2)


def attention(q, k, v, heads, scale, out=None, key_len=None):
    B, HD, N = q.shape
    D = HD // heads
    qh = q.reshape(B, heads, D, N)
    kh = k.reshape(B, heads, D, N)
    vh = v.reshape(B, heads, D, N)
    sim = torch.einsum("bhdn,bhdm->bhnm", qh, kh) * scale
    if key_len is not None:
        pad = torch.arange(N).view(1, 1, 1, N) >= key_len.view(B, 1, 1, 1)
        sim = sim.masked_fill(pad, float("-inf"))
    attn = sim.softmax(dim=-1)
    o = torch.einsum("bhnm,bhdm->bhdn", attn, vh).reshape(B, HD, N)
    if out is not None:
        out.copy_(o)
        return out
    return o


def lstm_bidir(G, whh_t, lengths=None, out=None):
    """Restates the bidirectional LSTM recurrence with packed-sequence semantics (PyTorch gate order i,f,g,o)."""
    B, R, N = G.shape
    H = R // 8
    Y = torch.zeros(B, 2 * H, N)
    for b in range(B):
        n = N if lengths is None else int(lengths[b])
        for d in range(2):
            Wt = whh_t[d]  # [H, 4H]
            h = torch.zeros(H)
            c = torch.zeros(H)
            order = range(n) if d == 0 else range(n - 1, -1, -1)
            for t in order:
                g = G[b, d * 4 * H:(d + 1) * 4 * H, t] + h @ Wt
                i, f, gg, o = torch.sigmoid(g[:H]), torch.sigmoid(g[H:2 * H]), torch.tanh(g[2 * H:3 * H]), \
                    torch.sigmoid(g[3 * H:])
                c = f * c + i * gg
                h = o * torch.tanh(c)
                Y[b, d * H:(d + 1) * H, t] = h
    if out is not None:
        out.copy_(Y)
        return out
    return Y


def colnorm_apply(x, stats, gamma, beta, *, gamma_plus_one=False, act=ACT_NONE, slope=0.0, lengths=None, out=None):
    n = (x - stats[:, :, 0].unsqueeze(1)) * stats[:, :, 1].unsqueeze(1)
    g = gamma.unsqueeze(-1)
    if gamma_plus_one:
        g = 1.0 + g
    y = n * g + beta.unsqueeze(-1)
    if act == ACT_LEAKY:
        y = F.leaky_relu(y, slope)
    if lengths is not None:
        L = x.shape[2]
        y = y.masked_fill((torch.arange(L).view(1, 1, L) >= lengths.view(-1, 1, 1)), 0.0)
    if out is not None:
        out.copy_(y)
        return out
    return y


def add_chanvec(x, v, out=None):
    y = x + v.unsqueeze(-1)
    if out is not None:
        out.copy_(y)
        return out
    return y


def mean_tokens(x, out=None, lengths=None):
    if lengths is None:
        m = x.mean(dim=2)
    else:
        m = torch.stack([x[b, :, :int(lengths[b])].mean(dim=1) for b in range(x.shape[0])])
    if out is not None:
        out.copy_(m)
        return out
    return m


def axpbypcz(x, a, y=None, b=0.0, z=None, c=0.0, out=None):
    r = a * x
    if y is not None:
        r = r + b * y
    if z is not None:
        r = r + c * z
    if out is not None:
        out.copy_(r)
        return out
    return r


def time_features(t, w, B, out=None):
    tt = torch.full((B, 1), float(t), dtype=torch.float32)
    freqs = tt * w.view(1, -1) * 2 * math.pi
    r = torch.cat([tt, freqs.sin(), freqs.cos()], dim=-1)
    if out is not None:
        out.copy_(r)
        return out
    return r


def tokens_to_channels(e, out, B=None):
    out.copy_(e.transpose(-1, -2) if e.dim() == 3 else e.t().unsqueeze(0).expand(out.shape[0], -1, -1))
    return out


def broadcast_cols(x, out):
    out.copy_(x.unsqueeze(-1).expand_as(out))
    return out


def copy_ncl(x, out):
    out.copy_(x)
    return out


def duration_head(x, w, bias, lengths=None, tail=0, want_sums=False):
    B, K, N = x.shape
    sums = torch.sigmoid(F.linear(x.transpose(1, 2), w, bias)).sum(dim=-1)
    dur = torch.round(sums).clamp(min=1).long()
    if lengths is not None:
        dur = dur.masked_fill(torch.arange(N).view(1, N) >= lengths.view(-1, 1), 0)
    last = (lengths.long() - 1) if lengths is not None else torch.full((B,), N - 1)
    dur[torch.arange(B), last] += tail
    return (dur, sums) if want_sums else dur


def expand_by_durations(x, dur, T, shift=False, out=None):
    B, C, N = x.shape
    idx = torch.stack([torch.repeat_interleave(torch.arange(N), dur[b], output_size=T) for b in range(B)])
    y = torch.gather(x, 2, idx.unsqueeze(1).expand(B, C, T))
    if shift:
        y = torch.cat([y[:, :, :1], y[:, :, :-1]], dim=2)
    if out is not None:
        out.copy_(y)
        return out
    return y


# ---- reference-audio style path (st2_style.hip contracts) ------------------------------------------------------------
def stft_frames(wave, n_win, hop, shift):
    B, L = wave.shape
    M = L // hop + 1
    idx = torch.arange(M).unsqueeze(0) * hop + torch.arange(n_win).unsqueeze(1) - shift   # [n_win, M]
    idx = idx.abs()
    idx = torch.where(idx >= L, 2 * (L - 1) - idx, idx)
    return wave[:, idx]


def power_spectrum(y):
    K = y.shape[1] // 2
    return y[:, :K] ** 2 + y[:, K:] ** 2


def log_norm_(x, eps, mean, std):
    x.copy_((torch.log(eps + x) - mean) / std)
    return x


def dwconv3x3s2(x, w, bias, out):
    B, H, Cc, Wd = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3), w.unsqueeze(1), bias, stride=2, padding=1, groups=Cc)   # [B, C, Ho, Wo]
    out.copy_(y.permute(0, 2, 1, 3))
    return out


def avgpool2x2(x, out):
    xc = x.permute(0, 2, 1, 3)
    if xc.shape[-1] % 2 != 0:
        xc = torch.cat([xc, xc[..., -1:]], dim=-1)
    out.copy_(F.avg_pool2d(xc, 2).permute(0, 2, 1, 3))
    return out
