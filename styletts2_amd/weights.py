"""Load-time weight transforms: weight-norm folding and packing into the kernels' layouts.

Done once per load_state_dict(); the reference instead re-evaluates weight-norm on every forward
(torch.nn.utils.weight_norm hook).
"""
import torch

from . import _hooks


def fold_weight_norm(g, v):
    """w = g * v / ||v||, norm over every dim but 0 (old-style torch weight_norm, dim=0).
    For ConvTranspose1d dim 0 is C_in (Modules/istftnet.py:319-322: `ups.*.weight_g` is (C_in,1,1))."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def pack_conv(w):
    """[C_out, C_in, ks] -> K-major [C_in*ks, w_ld] (row ci*ks+t, column co), w_ld = C_out rounded up to 4,
    zero padded; the layout `st2_conv1d` stages with 16-byte loads."""
    C_out, C_in, ks = w.shape
    w_ld = (C_out + 3) // 4 * 4
    wt = torch.zeros((C_in * ks, w_ld), dtype=torch.float32, device=w.device)
    wt[:, :C_out] = w.permute(1, 2, 0).reshape(C_in * ks, C_out)
    return wt.contiguous()


F16S_X_SCALE = 8.0  # power of two applied to the activated input before its hi/lo split (st2.h: x_scale)


def f16s_chunk(ks):
    """Input-channel padding granule of the split-f16 packing (== st2_conv1d_f16s_chunk)."""
    return 32 if ks <= 3 else 16


def f16s_co_block(C_out):
    """Output-channel padding granule of the split-f16 packing (== st2_conv1d_f16s_co_block)."""
    return 128 if C_out > 64 else (64 if C_out > 32 else 32)


class SplitConvWeight:
    """Conv weight pre-split for `st2_conv1d_f16s` / `st2_conv1d_xs` (include/st2.h): wq is a float16 tensor
    [C_in_pad/16, ks, 2, co_pad, 16] holding hi[0..7] | lo[0..7] of W[co] * w_scale[co] for 8 consecutive input
    channels; w_scale[co] is the power of two that puts max|W[co]| in [2^13, 2^14) -- per OUTPUT ROW, so rows whose
    magnitudes differ by many octaves (weight-norm gains of a trained checkpoint) all keep their lo halves in the
    normal f16 range.  `row_scale` = 1 / w_scale, float32 [co_pad] (the kernels' d.w_row_scale)."""

    def __init__(self, wq, row_scale, C_in, C_out, ks):
        self.wq, self.row_scale, self.C_in, self.C_out, self.ks = wq, row_scale, C_in, C_out, ks

    @property
    def cin_pad(self):
        return self.wq.shape[0] * 16

    @property
    def co_pad(self):
        return self.wq.shape[3]

    def to(self, device):
        return SplitConvWeight(self.wq.to(device), self.row_scale.to(device), self.C_in, self.C_out, self.ks)

    def dense(self):
        """fp32 [C_out, C_in, ks] value the packed halves represent: (hi + lo) / w_scale[co]."""
        n16, ks, _, co_pad, _ = self.wq.shape
        q = self.wq.float()
        w = q[..., :8] + q[..., 8:]                                 # [n16, ks, 2, co_pad, 8]
        w = w.permute(3, 0, 2, 4, 1).reshape(co_pad, n16 * 16, ks)  # [co, (n16, kg, e), t]
        w = w * self.row_scale.float().view(-1, 1, 1)
        return w[:self.C_out, :self.C_in].contiguous()


def pack_conv_f16s(w):
    """[C_out, C_in, ks] fp32 -> SplitConvWeight."""
    w = w.detach().float().cpu()
    C_out, C_in, ks = w.shape
    cin_pad = -(-C_in // f16s_chunk(ks)) * f16s_chunk(ks)
    co_pad = -(-C_out // f16s_co_block(C_out)) * f16s_co_block(C_out)
    amax = w.abs().reshape(C_out, -1).amax(dim=1)
    # exponent e of amax = m * 2^e, m in [0.5, 1)  ->  scale 2^(14 - e) puts amax in [2^13, 2^14); all-zero rows: 1.
    # The exponent is clamped at 126: a row whose amax is below 2^-112 (pruned / denormal weights) would otherwise get
    # scale = inf and poison the conv with inf * 0; such a row keeps its (negligible) values at scale 2^126 and a finite
    # 1 / scale.  Same rule in csrc/st2_engine.hip pack_split (the two packers are bit-identical).
    _, e = torch.frexp(amax)
    scale = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), torch.clamp(14 - e, max=126)), torch.ones_like(amax))
    ws = torch.zeros((co_pad, cin_pad, ks), dtype=torch.float32)
    ws[:C_out, :C_in] = w * scale.view(-1, 1, 1)
    hi = ws.half()
    lo = (ws - hi.float()).half()
    n16 = cin_pad // 16

    def arrange(h):  # [co, (n16, kg, e), t] -> [n16, t, kg, co, e]
        return h.reshape(co_pad, n16, 2, 8, ks).permute(1, 4, 2, 0, 3)

    wq = torch.cat([arrange(hi), arrange(lo)], dim=-1).contiguous()
    row_scale = torch.ones(co_pad, dtype=torch.float32)
    row_scale[:C_out] = 1.0 / scale
    return SplitConvWeight(wq, row_scale, C_in, C_out, ks)


def conv_precision():
    """Which kernel the convolutions are packed for: "f16s" (split-f16 MFMA, st2_conv1d_xs / st2_conv1d_f16s) or, from
    tests only (_hooks.py), "f32" (the exact-fp32 MFMA build of the same contract, st2_conv1d)."""
    return _hooks.conv_precision


def pack_conv_auto(w):
    """pack_conv_f16s() or pack_conv() according to conv_precision()."""
    return pack_conv_f16s(w) if conv_precision() == "f16s" else pack_conv(w)


def pack_linear(w):
    """nn.Linear weight [out, in] -> packed k=1 conv weight [in, out_ld]."""
    return pack_conv(w.unsqueeze(-1))


def pack_linear_auto(w):
    """nn.Linear weight [out, in] packed as a k=1 conv for the kernel conv_precision() selects."""
    return pack_conv_auto(w.unsqueeze(-1))


def polyphase_strided_conv(w, stride):
    """Strided Conv1d weight [C_out, C_in, K] with K == 2*stride -> the stride-1 Conv1d weight [C_out, C_in*stride, 2]
    that acts on the de-interleaved input  xp[ci*stride + r][u] = x[ci][u*stride + r - pad]  (st2_phase_split):
        y[co][q] = sum_{ci,r,j} w[co][ci][j*stride + r] * xp[ci*stride + r][q + j],  j in {0, 1}
    (noise_convs of both vocoders: Modules/istftnet.py:332-336, Modules/hifigan.py:296-300)."""
    C_out, C_in, K = w.shape
    assert K == 2 * stride, "polyphase form implemented for kernel = 2*stride (all reference configs)"
    return w.reshape(C_out, C_in, 2, stride).permute(0, 1, 3, 2).reshape(C_out, C_in * stride, 2).contiguous()


def polyphase_convt(w, stride):
    """ConvTranspose1d weight [C_in, C_out, K] with K == 2*stride -> equivalent Conv1d weight
    [stride*C_out, C_in, 2] (pad_left = 1, L_out = L_in + 1):
        Y[r*C_out + co][q] = sum_ci  w[ci,co,r+stride] * x[ci][q-1] + w[ci,co,r] * x[ci][q]
    and out[co][l] = Y[(l+pad) % stride][co][(l+pad) // stride]  (st2_convt_interleave)."""
    C_in, C_out, K = w.shape
    assert K == 2 * stride, "polyphase form implemented for kernel = 2*stride (all reference configs)"
    wp = torch.empty((stride, C_out, C_in, 2), dtype=w.dtype, device=w.device)
    for r in range(stride):
        wp[r, :, :, 0] = w[:, :, r + stride].t()
        wp[r, :, :, 1] = w[:, :, r].t()
    return wp.reshape(stride * C_out, C_in, 2)


def strip_module_prefix(state_dict):
    """Checkpoints saved from nn.DataParallel carry a `module.` prefix
    (Demo/Inference_LJSpeech.ipynb:199-215)."""
    out = {}
    for k, v in state_dict.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out
