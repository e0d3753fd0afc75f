"""Deterministic synthetic weights and inputs (there are no checkpoints or datasets offline).

`init_synthetic_` fills any module / state_dict with seeded values whose scale follows the reference's
initialisers (kaiming-uniform class for Linear/Conv, N(0,0.01)-class weight-normed convs, unit embeddings)
but with non-trivial weight-norm gains, Snake alphas, LayerNorm affines and biases so that every term of
every kernel is exercised.  Values depend only on (seed, tensor name, shape): the oracle, the engine and
the reference modules can be loaded with bit-identical weights on any machine with the same numpy.
"""
import math
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def synthetic_tensor(name, shape, seed, v_for_g=None):
    r = _rng(seed, name)
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    n = int(np.prod(shape)) if shape else 1

    def normal(std, mean=0.0):
        return (r.standard_normal(shape) * std + mean).astype(np.float32)

    def uniform(lo, hi):
        return r.uniform(lo, hi, size=shape).astype(np.float32)

    if leaf == "weight_g":
        assert v_for_g is not None
        norm = np.sqrt((v_for_g.reshape(shape[0], -1).astype(np.float64) ** 2).sum(axis=1)).reshape(shape)
        return (norm * r.uniform(0.7, 1.3, size=shape)).astype(np.float32)
    if leaf == "weight_v":
        fan_in = n // shape[0] if len(shape) > 1 else 1
        # ConvTranspose1d stores [C_in, C_out, K]; its fan-in is C_in*K/stride ~ C_in*2 -- use dim0*2
        return normal(1.0 / math.sqrt(3.0 * max(fan_in, 1)))
    if leaf.startswith("alpha") or ".alpha" in name or name.startswith("alpha") or ".alphas." in name:
        return uniform(0.6, 1.6)
    if leaf in ("gamma",) or (leaf == "weight" and len(shape) == 1):
        return normal(0.1, 1.0)
    if leaf in ("beta",):
        return normal(0.1)
    if "embedding" in name and leaf == "weight":
        return normal(1.0)
    if leaf == "weights":  # LearnedPositionalEmbedding, Modules/diffusion/modules.py:664
        return normal(1.0)
    if leaf.startswith("weight_ih") or leaf.startswith("weight_hh") or leaf.startswith("bias_ih") or \
            leaf.startswith("bias_hh"):
        hid = shape[0] // 4
        k = 1.0 / math.sqrt(hid)
        return uniform(-k, k)
    if leaf == "weight":
        fan_in = n // shape[0]
        return normal(1.0 / math.sqrt(3.0 * max(fan_in, 1)))
    if leaf == "bias":
        return normal(0.02)
    if leaf == "position_ids" or "int" in str(shape):
        return None
    return normal(0.05)


def synthetic_state_dict(template, seed):
    """template: a state_dict (name -> tensor) giving names/shapes/dtypes; returns a new CPU state_dict."""
    out = {}
    for name, t in template.items():
        if not torch.is_floating_point(t):
            out[name] = t.clone()
            continue
        if name.endswith("weight_g"):
            continue
        v = synthetic_tensor(name, t.shape, seed)
        out[name] = torch.from_numpy(v).reshape(t.shape)
    for name, t in template.items():
        if name.endswith("weight_g"):
            v = out[name[:-1] + "v"].numpy()
            out[name] = torch.from_numpy(synthetic_tensor(name, t.shape, seed, v_for_g=v)).reshape(t.shape)
    return {k: out[k] for k in template}


def init_synthetic_(module, seed):
    """In-place synthetic initialisation of an nn.Module (engine or reference)."""
    sd = synthetic_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return module


def init_trained_like_(module, seed, gain_sigma=0.7, alpha_decades=1.0):
    """`init_synthetic_`, then the two free parameters that decide how much of the f16 range a split-f16 conv operand uses
    in a TRAINED checkpoint (Modules/istftnet.py:27-62) get realistic spreads instead of their initial values: weight-norm
    gains `weight_g` are multiplied by log-normal factors exp(N(0, gain_sigma)) (trained gains differ by octaves between rows)
    and Snake `alpha` is drawn log-uniform in 10^[-alpha_decades, +alpha_decades] (the reference initialises it to 1)."""
    init_synthetic_(module, seed)
    sd = module.state_dict()
    out = {}
    for name, t in sd.items():
        if name.endswith("weight_g"):
            f = np.exp(_rng(seed, name + ".gain").standard_normal(tuple(t.shape)) * gain_sigma).astype(np.float32)
            out[name] = t * torch.from_numpy(f)
        elif any(part.startswith("alpha") for part in name.split(".")[-2:]):  # alpha1.0 / alpha2.2 / alpha (ParameterLists)
            u = _rng(seed, name + ".alpha").uniform(-alpha_decades, alpha_decades, tuple(t.shape)).astype(np.float32)
            out[name] = torch.from_numpy(10.0 ** u).reshape(t.shape)
        else:
            out[name] = t
    module.load_state_dict(out)
    return module


def scale_params_(module, rules):
    """Multiplies every parameter whose name contains a key of `rules` and ends in weight / weight_g / bias / gamma / beta by
    that key's factor (first match wins; `weight_v` is left alone: the gain carries a weight-normed layer's scale).  Tests use
    it to build checkpoints whose un-normalised conv inputs (GELU / LeakyReLU outputs, generator stage outputs) sit at 1e-2 ...
    1e-3 instead of O(1): a trained checkpoint is free to put them there, and the by-rule operand scale of the split-f16
    convs is then 5-500 x less precise than fp32 (pipeline.calibrate is what fixes it).  Returns the names it touched."""
    touched = []
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.split(".")[-1] not in ("weight", "weight_g", "bias", "gamma", "beta"):
                continue
            for key, f in rules.items():
                if key in name:
                    p.mul_(float(f))
                    touched.append(name)
                    break
    for m in module.modules():  # packed copies of the old values (engine-backed modules cache them)
        if hasattr(m, "refresh"):
            m.refresh()
    return touched


def init_spectral_norm_(module, seed):
    """Synthetic initialisation for modules under old-style spectral norm (`weight_orig` / `weight_u` / `weight_v`:
    the style encoders, models.py:97-164): seeded `weight_orig` / biases as in `init_synthetic_`, then u, v = the
    leading singular pair from a fixed number of fp64 power iterations, so that sigma = u.(W v) is the spectral norm
    and the folded weights are O(1) like a trained checkpoint's (random u, v would give an arbitrary, often tiny
    sigma)."""
    sd = module.state_dict()
    out = {}
    for name, t in sd.items():
        if name.endswith("weight_u") or name.endswith("weight_v") or not torch.is_floating_point(t):
            out[name] = t.clone()
            continue
        leaf = "weight" if name.endswith("weight_orig") else name.split(".")[-1]
        out[name] = torch.from_numpy(synthetic_tensor(name[:-len("weight_orig")] + leaf if name.endswith("weight_orig")
                                                      else name, t.shape, seed)).reshape(t.shape)
    for name in sd:
        if name.endswith("weight_orig"):
            w = out[name].double().reshape(out[name].shape[0], -1).numpy()
            r = _rng(seed, name + ".u")
            u = r.standard_normal(w.shape[0])
            for _ in range(30):
                v = w.T @ u
                v /= np.linalg.norm(v) + 1e-12
                u = w @ v
                u /= np.linalg.norm(u) + 1e-12
            base = name[:-len("weight_orig")]
            out[base + "weight_u"] = torch.from_numpy(u.astype(np.float32))
            out[base + "weight_v"] = torch.from_numpy(v.astype(np.float32))
    module.load_state_dict(out)
    return module


# ---- inputs --------------------------------------------------------------------------------------
def f0_contour(B, frames, seed):
    """Speech-like F0 (Hz) at the 2T frame rate: voiced arcs in 90-260 Hz with unvoiced (0 Hz) gaps."""
    r = np.random.default_rng([int(seed), 77])
    t = np.arange(frames, dtype=np.float64)
    out = np.zeros((B, frames), dtype=np.float32)
    for b in range(B):
        base = r.uniform(110, 220)
        f = base + 35.0 * np.sin(2 * np.pi * t / r.uniform(60, 140) + r.uniform(0, 6.28)) \
            + 12.0 * np.sin(2 * np.pi * t / r.uniform(9, 23)) + r.standard_normal(frames) * 1.5
        pos = 0
        while pos < frames:  # unvoiced gaps
            pos += int(r.integers(20, 70))
            gap = int(r.integers(3, 14))
            f[pos:pos + gap] = 0.0
            pos += gap
        out[b] = f.astype(np.float32)
    return torch.from_numpy(out)


def decoder_inputs(B, T, seed, hidden=512, style_dim=128, harmonics=9, samples_per_frame=300):
    """asr [B,hidden,T], F0_curve [B,2T], N [B,2T], s [B,style_dim], noise [B, 2T*300, 9]."""
    g = torch.Generator().manual_seed(int(seed))
    asr = torch.randn(B, hidden, T, generator=g)
    F0 = f0_contour(B, 2 * T, seed)
    N = torch.randn(B, 2 * T, generator=g).abs() * 0.5
    s = torch.randn(B, style_dim, generator=g)
    noise = torch.randn(B, 2 * T * samples_per_frame, harmonics, generator=g)
    return asr, F0, N, s, noise
