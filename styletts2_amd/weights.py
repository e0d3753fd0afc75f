"""Load-time weight transforms: weight-norm folding and packing into the kernels' layouts.

Done once per load_state_dict(); the reference instead re-evaluates weight-norm on every forward
(torch.nn.utils.weight_norm hook).
"""
import torch


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


def pack_linear(w):
    """nn.Linear weight [out, in] -> packed k=1 conv weight [in, out_ld]."""
    return pack_conv(w.unsqueeze(-1))


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
