"""TEST INFRASTRUCTURE: a CPU backend for the C++ launch plans (styletts2_amd/csrc/st2_engine.hip).

`st2_debug_set_backend` lets the caller replace every kernel / memory entry point the C++ plans call.  This module
provides the replacements as ctypes callbacks that decode the raw pointers + strides into torch CPU tensors and
evaluate the per-kernel contracts of oracle/ops_ref.py.  With it `st2_decoder_forward` / `st2_sampler_run` run on
HOST memory, so plan wiring, C++ weight packing and workspace aliasing are checked against the module-level oracle
without a GPU.  The HIP kernels themselves are held to the same contracts on the GPU box (tests/test_ops_gpu.py).
"""
import contextlib
import ctypes as C

import torch

from oracle import ops_ref as R
from styletts2_amd import _lib
from styletts2_amd.weights import SplitConvWeight

_keep = []  # host "device" allocations and callback objects must outlive the calls


def _t(ptr, shape, strides, dtype=torch.float32):
    """Strided tensor over raw memory (strides in elements)."""
    if not ptr:
        return None
    span = 1 + sum((s - 1) * abs(st) for s, st in zip(shape, strides))
    esz = torch.empty(0, dtype=dtype).element_size()
    buf = (C.c_char * (span * esz)).from_address(ptr)
    return torch.frombuffer(buf, dtype=dtype).as_strided(tuple(shape), tuple(strides))


def _ncl(ptr, bs, cs, B, Cc, L):
    return _t(ptr, (B, Cc, L), (bs, cs, 1))


def _gb(ptr, gb_bs, B, Cc):
    if not ptr:
        return None
    return _t(ptr, (1, Cc), (0, 1)) if gb_bs == 0 else _t(ptr, (B, Cc), (gb_bs, 1))


def _weight(d):
    n16 = d.wq_cin_pad // 16
    wq = _t(d.wq, (n16, d.ks, 2, d.wq_co_pad, 16), (d.ks * 2 * d.wq_co_pad * 16, 2 * d.wq_co_pad * 16, d.wq_co_pad * 16,
                                                     16, 1), torch.float16)
    rs = _t(d.w_row_scale, (d.wq_co_pad,), (1,))
    assert rs is not None, "the engine always passes per-row weight scales"
    return SplitConvWeight(wq, rs, d.C_in, d.C_out, d.ks)


def _epilogue_kwargs(d):
    kw = dict(dil=d.dil, pad_left=d.pad_left, L_out=d.L_out, bias=_t(d.bias, (d.C_out,), (1,)),
              out=_ncl(d.y, d.y_bs, d.y_cs, d.B, d.C_out, d.L_out), div=d.div, act=d.act, act_split=d.act_split,
              act_slope=d.act_slope, res_shift=d.res_shift)
    if d.res:
        kw["res"] = _ncl(d.res, d.res_bs, d.res_cs, d.B, d.C_out, (d.L_out + (1 << d.res_shift) - 1) >> d.res_shift)
    if d.res2:
        kw["res2"] = _ncl(d.res2, d.res2_bs, d.res2_cs, d.B, d.C_out, d.L_out)
    return kw


def _prologue_kwargs(pro, slope, stats, gamma, beta, gb_bs, gamma_plus_one, alpha, B, Cc, L, gb_seg=0):
    kw = dict(pro=pro, slope=slope, gamma_plus_one=bool(gamma_plus_one))
    if gb_seg:  # token-merged view: one affine row per segment of gb_seg columns
        assert pro == R.PRO_COLNORM and B == 1
        G = -(-L // gb_seg)
        kw.update(stats=_t(stats, (B, L, 2), (L * 2, 2, 1)), gamma=_t(gamma, (G, Cc), (gb_bs, 1)),
                  beta=_t(beta, (G, Cc), (gb_bs, 1)), gb_seg=gb_seg)
        return kw
    if pro in (R.PRO_ADAIN_LEAKY, R.PRO_ADAIN_SNAKE):
        kw.update(stats=_t(stats, (B, Cc, 2), (Cc * 2, 2, 1)), gamma=_gb(gamma, gb_bs, B, Cc), beta=_gb(beta, gb_bs, B, Cc))
    if pro == R.PRO_COLNORM:
        kw.update(stats=_t(stats, (B, L, 2), (L * 2, 2, 1)), gamma=_gb(gamma, gb_bs, B, Cc), beta=_gb(beta, gb_bs, B, Cc))
    if pro in (R.PRO_ADAIN_SNAKE, R.PRO_SNAKE):
        kw["alpha"] = _t(alpha, (Cc,), (1,))
    return kw


def _x_scale(pro):
    return 8.0 if pro in (R.PRO_ADAIN_LEAKY, R.PRO_ADAIN_SNAKE, R.PRO_COLNORM) else 1.0


X_SCALES_SEEN = []  # (pro, x_scale) of every split-f16 conv the plans issued: the rule's value, or the calibrated table's
EXPECT_RULE = [True]  # tests of st2_calibration_write switch the rule check off


def _check_x_scale(pro, x_scale):
    import math
    m, _ = math.frexp(x_scale)
    assert x_scale > 0 and m == 0.5, "x_scale must be a power of two, got %r" % (x_scale,)
    if EXPECT_RULE[0]:
        assert x_scale == _x_scale(pro)
    X_SCALES_SEEN.append((pro, x_scale))


def conv1d_f16s(dp, stream):
    d = dp.contents
    _check_x_scale(d.pro, d.x_scale)
    assert abs(d.out_scale * d.x_scale - 1.0) < 1e-12
    x = _ncl(d.x, d.x_bs, d.x_cs, d.B, d.C_in, d.L_in)
    kw = _epilogue_kwargs(d)
    kw.update(_prologue_kwargs(d.pro, d.slope, d.stats, d.gamma, d.beta, d.gb_bs, d.gamma_plus_one, d.alpha, d.B, d.C_in,
                               d.L_in))
    y = R._conv1d(x, _weight(d), d.C_out, d.ks, **kw)
    _emit_part(d, y)
    return 0


def _emit_part(d, y):
    """Per-slot shifted (sum, sum of squares) of the stored output, the contract of d.part (st2.h)."""
    if d.part:
        nt = d.part_nt
        cols = getattr(d, "part_cols", 0) or 128  # st2.h: 0 = 128 columns per slot; 64 / 32 on small grids (xs only)
        assert cols in (128, 64, 32) and nt * cols >= d.L_out
        _shifted_sums(d.part, y, nt, cols)


def act_split(x, x_bs, x_cs, B, Cc, L, pro, slope, stats, gamma, beta, gb_bs, gb_seg, gamma_plus_one, alpha, x_scale, xs,
              xs_cg, Lp, halo, stream):
    _check_x_scale(pro, x_scale)
    xv = _ncl(x, x_bs, x_cs, B, Cc, L)
    kw = _prologue_kwargs(pro, slope, stats, gamma, beta, gb_bs, gamma_plus_one, alpha, B, Cc, L, gb_seg)
    u = R.activate(xv, **kw) * x_scale
    hi = u.half()
    lo = (u - hi.float()).half()
    planes = _t(xs, (B, 2, xs_cg, Lp, 8), (2 * xs_cg * Lp * 8, xs_cg * Lp * 8, Lp * 8, 8, 1), torch.float16)
    planes.zero_()
    for p, v in ((0, hi), (1, lo)):
        full = torch.zeros(B, xs_cg * 8, L, dtype=torch.float16)
        full[:, :Cc] = v
        planes[:, p, :, halo:halo + L, :] = full.reshape(B, xs_cg, 8, L).permute(0, 1, 3, 2)
    return 0


def conv1d_xs(dp, stream):
    d = dp.contents
    assert d.pad_left <= d.xs_halo and abs(d.out_scale * d.x_scale - 1.0) < 1e-12
    planes = _t(d.xs, (d.B, 2, d.xs_cg, d.xs_lp, 8), (2 * d.xs_cg * d.xs_lp * 8, d.xs_cg * d.xs_lp * 8, d.xs_lp * 8, 8, 1),
                torch.float16)
    u = (planes[:, 0].float() + planes[:, 1].float()) / d.x_scale                 # [B, cg, Lp, 8]
    u = u.permute(0, 1, 3, 2).reshape(d.B, d.xs_cg * 8, d.xs_lp)[:, :d.C_in, d.xs_halo:d.xs_halo + d.L_in]
    assert float(planes[:, :, :, :d.xs_halo].float().abs().max()) == 0.0, "halo must be zero"
    kw = _epilogue_kwargs(d)
    y = R._conv1d(u.contiguous(), _weight(d), d.C_out, d.ks, **kw)
    _emit_part(d, y)
    return 0


def _shifted_sums(part_ptr, y, nt, cols):
    """The contract of every `part` output (st2.h, ABI v20): float2 [B * C][nt] = (sum, sum of squares) of (y - shift) over each
    slot's columns, then float [B * C][nt] = the shifts, each slot's first stored value y[b, c, i * cols]."""
    B, Cc, L = y.shape
    part = _t(part_ptr, (B, Cc, nt, 2), (Cc * nt * 2, nt * 2, 2, 1))
    shifts = _t(part_ptr + B * Cc * nt * 2 * 4, (B, Cc, nt), (Cc * nt, nt, 1))
    yd = torch.zeros(B, Cc, nt * cols, dtype=torch.float64)
    yd[:, :, :L] = y.double()
    yd = yd.reshape(B, Cc, nt, cols)
    valid = (torch.arange(nt * cols) < L).reshape(nt, cols)
    dv = (yd - yd[..., :1]) * valid
    part[..., 0] = dv.sum(-1).float()
    part[..., 1] = (dv * dv).sum(-1).float()
    shifts.copy_(yd[..., 0].float())


def stats_finalize(part, rows, nt, L, eps, stats, cols, stream):
    """Chan's combination of the shifted per-slot sums in fp64 (== stats_finalize_kernel)."""
    p = _t(part, (rows, nt, 2), (nt * 2, 2, 1)).double()
    shift = _t(part + rows * nt * 2 * 4, (rows, nt), (nt, 1)).double()
    ns = min(nt, -(-L // cols))
    n = torch.tensor([min(cols, L - i * cols) for i in range(ns)], dtype=torch.float64)
    s1, s2 = p[:, :ns, 0], p[:, :ns, 1]
    mi = shift[:, :ns] + s1 / n
    mean = (n * mi).sum(-1) / L
    m2 = ((s2 - s1 * s1 / n) + n * (mi - mean.unsqueeze(-1)) ** 2).sum(-1)
    var = (m2 / L).clamp(min=0.0)
    st = _t(stats, (rows, 2), (2, 1))
    st[:, 0] = mean.float()
    st[:, 1] = (1.0 / torch.sqrt(var + eps)).float()
    return 0


def conv1d_direct(x, x_bs, x_cs, w, bias, y, y_bs, y_cs, B, C_in, C_out, L_in, L_out, ks, stride, pad, stream):
    R.conv1d_direct(_ncl(x, x_bs, x_cs, B, C_in, L_in), _t(w, (C_out, C_in, ks), (C_in * ks, ks, 1)),
                    _t(bias, (C_out,), (1,)), stride, pad, L_out=L_out, out=_ncl(y, y_bs, y_cs, B, C_out, L_out))
    return 0


def phase_split(x, x_bs, x_cs, B, Cc, L_in, stride, pad, xp, p_bs, p_cs, Lu, stream):
    _ncl(xp, p_bs, p_cs, B, Cc * stride, Lu).copy_(R.phase_split(_ncl(x, x_bs, x_cs, B, Cc, L_in), stride, pad, Lu))
    return 0


def instnorm_stats(x, x_bs, x_cs, B, Cc, L, eps, stats, stream):
    R.instnorm_stats(_ncl(x, x_bs, x_cs, B, Cc, L), eps, out=_t(stats, (B, Cc, 2), (Cc * 2, 2, 1)))
    return 0


def colnorm_stats(x, x_bs, x_cs, B, Cc, L, eps, stats, stream):
    R.colnorm_stats(_ncl(x, x_bs, x_cs, B, Cc, L), eps, out=_t(stats, (B, L, 2), (L * 2, 2, 1)))
    return 0


def style_fc(s, B, K, wt, bias, J, act, h, stream):
    R.style_fc(_t(s, (B, K), (K, 1)), _t(wt, (K, J), (J, 1)), _t(bias, (J,), (1,)), act, out=_t(h, (B, J), (J, 1)))
    return 0


def convt_interleave_stats(ph, p_bs, p_cs, Lq, bias, add, a_bs, a_cs, out, o_bs, o_cs, B, Cc, stride, pad, L_raw,
                           reflect_left, part, part_nt, stream):
    L_out = L_raw + reflect_left
    y = R._convt_interleave(_ncl(ph, p_bs, p_cs, B, stride * Cc, Lq), Cc, stride, pad, L_raw,
                            bias=_t(bias, (Cc,), (1,)), add=_ncl(add, a_bs, a_cs, B, Cc, L_out) if add else None,
                            reflect_left=bool(reflect_left), out=_ncl(out, o_bs, o_cs, B, Cc, L_out))
    if part:
        _shifted_sums(part, y, part_nt, 1024)
    return 0


def adain_leaky_pool(x, x_bs, x_cs, stats, gamma, beta, gb_bs, slope, w, bias, y, y_bs, y_cs, B, Cc, L, stream):
    R.adain_leaky_pool(_ncl(x, x_bs, x_cs, B, Cc, L), _t(stats, (B, Cc, 2), (Cc * 2, 2, 1)), _gb(gamma, gb_bs, B, Cc).expand(B, Cc),
                       _gb(beta, gb_bs, B, Cc).expand(B, Cc), slope, _t(w, (Cc, 3), (3, 1)), _t(bias, (Cc,), (1,)),
                       out=_ncl(y, y_bs, y_cs, B, Cc, 2 * L))
    return 0


def har_source(f0, B, Fr, U, H, noise, lin_w, lin_b, sine_amp, noise_std, vthr, sr, scratch, out, stream):
    y = R.har_source(_t(f0, (B, Fr), (Fr, 1)), U, _t(noise, (B, Fr * U, H), (Fr * U * H, H, 1)), _t(lin_w, (H,), (1,)),
                     _t(lin_b, (1,), (1,)), sine_amp=sine_amp, noise_std=noise_std, voiced_threshold=vthr, sample_rate=sr)
    _t(out, (B, Fr * U), (Fr * U, 1)).copy_(y)
    return 0


def stft_mag_phase(x, B, L, n_fft, hop, har, har_bs, har_cs, stream):
    _ncl(har, har_bs, har_cs, B, n_fft + 2, L // hop + 1).copy_(R.stft_mag_phase(_t(x, (B, L), (L, 1)), n_fft, hop))
    return 0


def istft(sp, sp_bs, sp_cs, B, M, n_fft, hop, wave, wave_bs, stream):
    y = R.istft(_ncl(sp, sp_bs, sp_cs, B, n_fft + 2, M), n_fft, hop)
    _t(wave, (B, hop * (M - 1)), (wave_bs, 1)).copy_(y.reshape(B, -1))
    return 0


def attention_keylen(q, k, v, bs, cs, o, o_bs, o_cs, B, H, D, N, scale, key_len, stream):
    kl = _t(key_len, (B,), (1,), torch.int32) if key_len else None
    R.attention(_ncl(q, bs, cs, B, H * D, N), _ncl(k, bs, cs, B, H * D, N), _ncl(v, bs, cs, B, H * D, N), H, scale,
                out=_ncl(o, o_bs, o_cs, B, H * D, N), key_len=kl)
    return 0


def add_chanvec(x, x_bs, x_cs, v, v_bs, y, y_bs, y_cs, B, Cc, N, stream):
    R.add_chanvec(_ncl(x, x_bs, x_cs, B, Cc, N), _t(v, (B, Cc), (v_bs, 1)), out=_ncl(y, y_bs, y_cs, B, Cc, N))
    return 0


def mean_tokens_len(x, x_bs, x_cs, m, m_bs, B, Cc, N, length, stream):
    ln = _t(length, (B,), (1,), torch.int32) if length else None
    R.mean_tokens(_ncl(x, x_bs, x_cs, B, Cc, N), out=_t(m, (B, Cc), (m_bs, 1)), lengths=ln)
    return 0


def axpbypcz(x, a, y, b, z, c, out, n, stream):
    R.axpbypcz(_t(x, (n,), (1,)), a, _t(y, (n,), (1,)) if y else None, b, _t(z, (n,), (1,)) if z else None, c,
               out=_t(out, (n,), (1,)))
    return 0


def time_features(t, w, H2, B, out, stream):
    R.time_features(t, _t(w, (H2,), (1,)), B, out=_t(out, (B, 1 + 2 * H2), (1 + 2 * H2, 1)))
    return 0


def tokens_to_channels(e, e_bs, B, N, E, y, y_bs, y_cs, stream):
    src = _t(e, (B, N, E), (e_bs, E, 1))
    _ncl(y, y_bs, y_cs, B, E, N).copy_(src.transpose(1, 2))
    return 0


def broadcast_cols(x, x_bs, y, y_bs, y_cs, B, Cc, N, stream):
    R.broadcast_cols(_t(x, (B, Cc), (x_bs, 1)), _ncl(y, y_bs, y_cs, B, Cc, N))
    return 0


def copy_ncl(x, x_bs, x_cs, y, y_bs, y_cs, B, Cc, L, stream):
    _ncl(y, y_bs, y_cs, B, Cc, L).copy_(_ncl(x, x_bs, x_cs, B, Cc, L))
    return 0


def expand_by_durations(x, x_bs, x_cs, dur, B, Cc, N, T, shift, y, y_bs, y_cs, stream):
    d = _t(dur, (B, N), (N, 1), torch.int64)
    _ncl(y, y_bs, y_cs, B, Cc, T).copy_(R.expand_by_durations(_ncl(x, x_bs, x_cs, B, Cc, N), d, T, shift=bool(shift)))
    return 0


def lstm_bidir(G, g_bs, g_cs, whh_t, lengths, B, H, N, Y, y_bs, y_cs, scratch, scratch_bytes, stream):
    lens = _t(lengths, (B,), (1,), torch.int32) if lengths else None
    R.lstm_bidir(_ncl(G, g_bs, g_cs, B, 8 * H, N), _t(whh_t, (2, H, 4 * H), (H * 4 * H, 4 * H, 1)), lens,
                 out=_ncl(Y, y_bs, y_cs, B, 2 * H, N))
    return 0


def colnorm_apply(x, x_bs, x_cs, stats, gamma, beta, gb_bs, gamma_plus_one, act, slope, length, y, y_bs, y_cs, B, Cc, L,
                  stream):
    lens = _t(length, (B,), (1,), torch.int32) if length else None
    R.colnorm_apply(_ncl(x, x_bs, x_cs, B, Cc, L), _t(stats, (B, L, 2), (L * 2, 2, 1)), _gb(gamma, gb_bs, B, Cc),
                    _gb(beta, gb_bs, B, Cc), gamma_plus_one=bool(gamma_plus_one), act=act, slope=slope, lengths=lens,
                    out=_ncl(y, y_bs, y_cs, B, Cc, L))
    return 0


def duration_head(x, x_bs, x_cs, w, bias, B, K, J, N, length, tail, dur, dsum, stream):
    lens = _t(length, (B,), (1,), torch.int32) if length else None
    d, sums = R.duration_head(_ncl(x, x_bs, x_cs, B, K, N), _t(w, (J, K), (K, 1)), _t(bias, (J,), (1,)), lengths=lens,
                              tail=tail, want_sums=True)
    _t(dur, (B, N), (N, 1), torch.int64).copy_(d)
    if dsum:
        _t(dsum, (B, N), (N, 1)).copy_(sums)
    return 0


def mask_tail(x, x_bs, x_cs, B, Cc, L, length, stream):
    lens = _t(length, (B,), (1,), torch.int32)
    xv = _ncl(x, x_bs, x_cs, B, Cc, L)
    xv.masked_fill_(torch.arange(L).view(1, 1, L) >= lens.view(-1, 1, 1), 0.0)
    return 0


def embed_tokens(tokens, B, N, table, V, E, add, pos, length, y, y_bs, y_cs, stream):
    tok = _t(tokens, (B, N), (N, 1), torch.int64)
    valid = (tok >= 0) & (tok < V)                                           # st2.h: ids outside [0, V) contribute 0
    emb = _t(table, (V, E), (E, 1))[tok.clamp(0, V - 1)] * valid.unsqueeze(-1)  # [B, N, E]
    if add:
        emb = emb + _t(add, (E,), (1,))
    if pos:
        emb = emb + _t(pos, (N, E), (E, 1)).unsqueeze(0)
    emb = emb.transpose(1, 2)                                                # [B, E, N]
    if length:
        lens = _t(length, (B,), (1,), torch.int32)
        emb = emb.masked_fill(torch.arange(N).view(1, 1, N) >= lens.view(-1, 1, 1), 0.0)
    _ncl(y, y_bs, y_cs, B, E, N).copy_(emb)
    return 0


def dwconv3x3s2(x, x_bs, x_hs, x_cs, w, bias, B, Cc, H, W, y, y_bs, y_hs, y_cs, stream):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    R.dwconv3x3s2(_t(x, (B, H, Cc, W), (x_bs, x_hs, x_cs, 1)), _t(w, (Cc, 3, 3), (9, 3, 1)), _t(bias, (Cc,), (1,)),
                  _t(y, (B, Ho, Cc, Wo), (y_bs, y_hs, y_cs, 1)))
    return 0


def avgpool2x2(x, x_bs, x_hs, x_cs, B, Cc, H, W, y, y_bs, y_hs, y_cs, stream):
    R.avgpool2x2(_t(x, (B, H, Cc, W), (x_bs, x_hs, x_cs, 1)), _t(y, (B, H // 2, Cc, (W + 1) // 2), (y_bs, y_hs, y_cs, 1)))
    return 0


def dev_alloc(nbytes):
    buf = C.create_string_buffer(int(nbytes) + 512)
    addr = (C.addressof(buf) + 255) & ~255
    _keep.append(buf)
    return addr


def dev_free(ptr):
    return None


def upload(dst, src, nbytes):
    C.memmove(dst, src, nbytes)
    return 0


_MEM_TYPES = {"dev_alloc": C.CFUNCTYPE(C.c_void_p, C.c_int64), "dev_free": C.CFUNCTYPE(None, C.c_void_p),
              "upload": C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)}
_p, _i64, _i32 = C.c_void_p, C.c_int64, C.c_int32
# slots whose signature is not that of the C-ABI entry point of the same name
_SPECIAL_TYPES = {"lstm_bidir": C.CFUNCTYPE(C.c_int, _p, _i64, _i32, _p, _p, _i32, _i32, _i32, _p, _i64, _i32, _p, _i64, _p)}


def _guard(fn):
    def run(*a):
        try:
            with torch.no_grad():
                return fn(*a)
        except Exception:  # an exception must not cross the C frames: report and fail the call
            import traceback
            traceback.print_exc()
            return 1
    return run


def install():
    """Builds the callback table and installs it; returns an object that must be kept alive while it is in use."""
    lib = _lib.load()
    table = (C.c_void_p * len(_lib.BACKEND_SLOTS))()
    cbs = []
    for i, name in enumerate(_lib.BACKEND_SLOTS):
        if name in _MEM_TYPES:
            cb = _MEM_TYPES[name](globals()[name])
        elif name in _SPECIAL_TYPES:
            cb = _SPECIAL_TYPES[name](_guard(globals()[name]))
        else:
            res, args = _lib._SIGNATURES["st2_" + name]
            cb = C.CFUNCTYPE(res, *args)(_guard(globals()[name]))
        cbs.append(cb)
        table[i] = C.cast(cb, C.c_void_p)
    _lib.check(lib.st2_debug_set_backend(table, len(_lib.BACKEND_SLOTS)), "st2_debug_set_backend")
    return cbs, table


@contextlib.contextmanager
def cpu_backend():
    keep = install()
    try:
        yield keep
    finally:
        _lib.load().st2_debug_set_backend(None, 0)
