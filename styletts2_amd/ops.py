"""Tensor-level wrappers over the C ABI (include/st2.h).

Every function takes PyTorch-ROCm tensors (fp32, device memory, unit stride along the last axis),
forwards raw pointers + strides to libst2_hip.so on torch's current HIP stream and returns torch
tensors.  PyTorch is plumbing here (allocation, streams); all arithmetic happens in the HIP kernels.
There is no CPU path: a tensor that is not on a HIP device raises.
"""
import ctypes as C

import torch

from . import _hooks, _lib
from .weights import F16S_X_SCALE, SplitConvWeight
from ._lib import (ACT_EXP_SIN, ACT_GELU, ACT_GELU_TANH, ACT_LEAKY, ACT_NONE, ACT_TANH, PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE,  # noqa: F401
                   PRO_COLNORM, PRO_LEAKY, PRO_NONE, PRO_SNAKE, ConvDesc)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_AUX_STREAMS = {}


def aux_stream(dev, priority=0, index=0):
    """ONE auxiliary stream per (device, priority, index), created on first use and reused by everything that needs "a second stream"
    (graph-capture warm-ups, the long-form front).  HIP maps its streams onto a handful of hardware queues (4 by default); a
    process that keeps asking torch for new streams walks through torch's pool and sooner or later gets one that shares the
    hardware queue of the stream it is meant to overlap with -- the two then serialise (measured: the second model of a process
    138 ms two-stream against 118 for the first; profiles/LAB_NOTES.md round 5).  `index` > 0: further streams of the same kind
    (the long-form decoders of independent sentences, pipeline.synthesize_long decode_streams)."""
    dev = torch.device(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(priority), int(index))
    if key not in _AUX_STREAMS:
        _AUX_STREAMS[key] = torch.cuda.Stream(dev, priority=int(priority))
    return _AUX_STREAMS[key]


_PROBE_BUF = {}


def wait_blocks(waiter, victim, spin_cycles=3_000_000, mode="wait"):
    """Does an event wait queued on `waiter` (mode "wait") -- or a long kernel running on it (mode "kernel") -- hold up kernels
    of `victim`?  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default): a `hipStreamWaitEvent`
    is a barrier packet at the head of its stream's HARDWARE queue -- until the event fires nothing behind it in that queue runs,
    whatever stream it belongs to.  Probe: a spin kernel (~1.5 ms) on a stream of its own, `waiter` waits for it (or runs the
    spin itself), a tiny kernel goes to `victim`; blocked iff the tiny kernel finishes only after the spin.  Both streams are
    warmed first (the first launch on a stream creates its queue).  Synchronises the device; start-up use only."""
    dev = victim.device
    key = dev.index
    if key not in _PROBE_BUF:
        _PROBE_BUF[key] = (torch.zeros(64, device=dev), torch.cuda.Stream(dev))
    buf, spin = _PROBE_BUF[key]
    for st in (waiter, victim, spin):
        with torch.cuda.stream(st):
            buf.add_(0.0)
    torch.cuda.synchronize(dev)
    e_spin, e_tiny = torch.cuda.Event(), torch.cuda.Event()
    src = waiter if mode == "kernel" else spin
    with torch.cuda.stream(src):
        torch.cuda._sleep(int(spin_cycles))
        e_spin.record(src)
    if mode != "kernel":
        waiter.wait_event(e_spin)
    with torch.cuda.stream(victim):
        buf.add_(1.0)
        e_tiny.record(victim)
    while not e_tiny.query():
        pass
    blocked = e_spin.query()  # the spin was over before the tiny kernel's completion was seen: it sat behind it
    torch.cuda.synchronize(dev)
    return bool(blocked)


def _chk(t, name, ndim=None):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.St2Error("%s must live on a HIP device (got %s); the engine has no CPU path" % (name, t.device))
    if t.dtype != torch.float32:
        raise _lib.St2Error("%s must be float32 (got %s)" % (name, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise _lib.St2Error("%s must be %d-D (got shape %s)" % (name, ndim, tuple(t.shape)))
    if t.dim() > 0 and t.shape[-1] > 1 and t.stride(-1) != 1:
        raise _lib.St2Error("%s must have unit stride along its last axis" % name)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _bs_cs(t):
    """(batch stride, channel stride) of an NCL view."""
    return t.stride(0), t.stride(1)


def x_scale_for(pro):
    """Power of two applied to the activated conv input before its f16 hi/lo split.  Normalised inputs (AdaIN /
    LayerNorm prologues: O(1) values) take 8, which keeps the lo halves of typical activations in the normal f16
    range; un-normalised inputs (plain / LeakyReLU / Snake prologues: the decoder's `cat` buffer carries the F0 curve
    in Hz, generator stage outputs, FFN intermediates) take 1, i.e. the full +-65504 of f16 (the lo half is then exact
    to 2^-25 absolute through f16 subnormals).  Beyond the range the kernels clamp and raise STATUS_F16_RANGE."""
    return F16S_X_SCALE if pro in (PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE, PRO_COLNORM) else 1.0


def calibrated_x_scale(max_abs, margin_bits=3):
    """`st2_calibration_scale`: the power of two that puts max |pro(x)| = `max_abs` into the top octave below 2^(16 -
    margin_bits) of the f16 range -- what `Engine.calibrate` installs per conv site (0.0 for max_abs <= 0: keep the rule)."""
    return float(_lib.load().st2_calibration_scale(float(max_abs), int(margin_bits)))


def status(clear=False):
    """The library's sticky device-side status word (include/st2.h `st2_status`): no synchronisation; a bit is visible
    once the kernel that raised it has completed."""
    return _lib.load().st2_status(1 if clear else 0)


lstm_recoveries = 0  # check_status() calls that found ST2_STATUS_LSTM_RECOVERED (bench.py reports it)


def check_status(ignore=0):
    """Raises St2Error if a kernel reported a device-side condition since the last check (and clears it).  Called by
    the pipeline at its existing host synchronisation points and at the start of every call for the previous one's
    kernels, so a failure is never silent and costs no extra synchronisation.  `ignore`: status bits the caller expects
    (pipeline.calibrate provokes F16_RANGE on purpose); every other bit is still reported."""
    st = status(clear=True)
    if st > 0:
        st &= ~int(ignore)
    if st > 0 and st & _lib.STATUS_LSTM_RECOVERED:  # informational: the outputs are valid, the call lost its latency advantage
        global lstm_recoveries
        lstm_recoveries += 1
        import warnings
        warnings.warn("a cooperative BiLSTM group was not co-resident in time; the call was re-run on the single-CU kernel "
                      "in-stream (results valid; ST2_STATUS_LSTM_RECOVERED)", RuntimeWarning, stacklevel=2)
        st &= ~_lib.STATUS_LSTM_RECOVERED
    if st <= 0:
        return
    msgs = []
    if st & _lib.STATUS_F16_RANGE:
        msgs.append("a split-f16 conv operand exceeded the f16 range (|x * x_scale| > 65504) and was clamped: the "
                    "result is finite but wrong; calibrate the operand scales for this checkpoint (pipeline.calibrate / st2_calibrate) and "
                    "check it with tools/validate_checkpoint.py, which names the conv site and its headroom")
    if st & _lib.STATUS_LSTM_TIMEOUT:
        msgs.append("a cooperative BiLSTM group timed out (its workgroups were not co-resident in time) on a launch without "
                    "the recovery pass: outputs of that call are invalid")
    if st & _lib.STATUS_DURATION_SUM:
        msgs.append("a row of the supplied durations does not sum to the frame count given with it (`total_frames`): the "
                    "alignment of that call repeated its last phoneme")
    raise _lib.St2Error("device-side status 0x%x: %s" % (st, "; ".join(msgs)))


class conv_autotune:
    """`with ops.conv_autotune(): forward(...)` -- start-up autotuning of the xs convs (include/st2.h `st2_conv_tune`): the
    first launch of every shape class inside the block times its bitwise-equivalent builds (tile shape / occupancy,
    dispatch-order or XCD-aware tile order) on this box and keeps the fastest; later calls (inside or outside the block, eager or graph-captured) run it.  Boxes of the same SKU differ by
    up to 1.75 x on individual classes with the rule's build, so a serving process runs one forward per batch shape in
    here before taking traffic.  `reset=True` forgets earlier measurements of the current device first."""

    def __init__(self, reset=False):
        self.reset = reset

    def __enter__(self):
        lib = _lib.load()
        if self.reset:
            _lib.check(lib.st2_conv_tune(-1), "st2_conv_tune")
        _lib.check(lib.st2_conv_tune(1), "st2_conv_tune")
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        _lib.check(_lib.load().st2_conv_tune(0), "st2_conv_tune")
        return False


TUNE_VARIANT_BITS = {1: "128x256 tiles, 2 wg/CU", 2: "XCD-aware tile order"}


def tune_variant_name(v):
    if v < 0:
        return "rule"
    names = [n for b, n in TUNE_VARIANT_BITS.items() if v & b]
    return " + ".join(names) if names else "128x128 tiles, 3 wg/CU, dispatch order"


def conv_tune_table():
    """The autotuner's table for the current device: a list of dicts {ks, C_in, C_out, L, B, chosen, candidates: [{variant,
    name, ms}]} (ms = 0 for classes pinned by hand)."""
    import ctypes
    lib = _lib.load()
    n = lib.st2_conv_tune_read(None, 0)
    rows = (ctypes.c_double * (24 * max(n, 1)))()
    n = min(n, lib.st2_conv_tune_read(rows, n))
    out = []
    for i in range(n):
        r = rows[24 * i:24 * i + 24]
        cands = [{"variant": int(r[8 + 2 * j]), "name": tune_variant_name(int(r[8 + 2 * j])), "ms": round(r[9 + 2 * j], 5)}
                 for j in range(int(r[7]))]
        out.append({"ks": int(r[0]), "C_in": int(r[1]), "C_out": int(r[2]), "L": int(r[3]), "B": int(r[4]),
                    "chosen": int(r[6]), "chosen_name": tune_variant_name(int(r[6])), "candidates": cands})
    return out


def conv_tune_set(ks, C_in, C_out, L_out, B, variant):
    """Pins (variant >= 0) or erases (-1) the build of one shape class on the current device (tests, A/B probes)."""
    _lib.check(_lib.load().st2_conv_tune_set(ks, C_in, C_out, L_out, B, variant), "st2_conv_tune_set")


def probe_cu_health():
    """(report dict, CU mask as a list of 32-bit words, number of excluded CUs) -- include/st2.h `st2_probe_cu_health`: which
    CUs of this box run the conv path's workgroups abnormally slowly, and the CU mask of the device without them."""
    import ctypes
    import json
    buf = ctypes.create_string_buffer(16384)
    mask = (ctypes.c_uint32 * 16)()
    n = ctypes.c_int32(0)
    _lib.check(_lib.load().st2_probe_cu_health(buf, len(buf), mask, 16, ctypes.byref(n)), "st2_probe_cu_health")
    rep = json.loads(buf.value.decode())
    words = (rep["cus"] + 31) // 32
    return rep, [int(mask[i]) for i in range(words)], int(n.value)


class headroom:
    """`with ops.headroom() as h: forward(...)` then `h.rows`: where every split-f16 conv operand of that forward sat in the
    f16 range, BOTH ends (include/st2.h `st2_debug_headroom`; debug hook: extra launches, a scratch allocation, synchronises).
    rows = [{index, kind ("act_split" | "fused conv"), pro, B, C, L, x_scale, max_abs, frac, rel_err, sub_share, site}]:
    frac = max |x_scale * pro(x)| / 65504 (>= 1: clamped, ST2_STATUS_F16_RANGE); rel_err = relative RMS error the hi/lo split
    adds to the operand (fp32 storage: 3.4e-8; all lo halves normal: ~4e-8; an operand at 1e-3 with x_scale 1: ~2e-5);
    sub_share = share of the operand's energy in elements whose lo half is a subnormal f16; site = the engine's conv site
    (Engine.calibration() index) or -1 for per-kernel calls."""

    def __enter__(self):
        _lib.check(_lib.load().st2_debug_headroom(1), "st2_debug_headroom")
        self.rows = []
        return self

    def __exit__(self, *exc):
        import ctypes
        lib = _lib.load()
        torch.cuda.synchronize()
        _lib.check(lib.st2_debug_headroom(0), "st2_debug_headroom")
        n = lib.st2_debug_headroom_read(None, 0)
        W = _lib.HEADROOM_COLS
        buf = (ctypes.c_double * (W * max(n, 1)))()
        n = min(n, lib.st2_debug_headroom_read(buf, n))
        pro_names = ["none", "leaky", "adain+leaky", "adain+snake", "snake", "layernorm"]
        for i in range(max(n, 0)):
            r = buf[W * i:W * i + W]
            self.rows.append({"index": i, "kind": "fused conv" if int(r[0]) else "act_split", "pro": pro_names[int(r[1])],
                              "B": int(r[2]), "C": int(r[3]), "L": int(r[4]), "x_scale": r[5], "max_abs": r[6], "frac": r[7],
                              "rel_err": r[8], "sub_share": r[9], "site": int(r[10]), "engine": int(r[11])})
        return False


def probe_box(level=0):
    """Micro-measurements of the current device as a dict (include/st2.h `st2_probe_box`; ~0.5 s, synchronises)."""
    import ctypes
    import json
    buf = ctypes.create_string_buffer(16384)
    _lib.check(_lib.load().st2_probe_box(buf, len(buf), level), "st2_probe_box")
    return json.loads(buf.value.decode())


XS_HALO = 32  # zero columns in front of every xs row (>= the largest pad_left on the path: 25)
FUSED_MAX_C, FUSED_K3_MAX_C = 64, 128  # see prefer_fused()
XS_MIN_L = 256  # shorter rows stay on the fused kernel: an xs row is >= 640 slots
XS_MIN_C_PLAIN = 64  # prologue-free convs take the xs pair too from this many input channels on (the split pass is
#                      then cheap next to the conv; measured 39 us vs 80 us per denoiser Linear at B*N = 3200)


def prefer_fused(pro, C_in, ks):
    """Layers whose xs pair is HBM-bound take the fused kernel instead (prologue arithmetic on the VALU beside the MFMAs,
    no activation pass: 12 instead of 20 bytes per element): the narrow HiFi-GAN stages (C <= 64) and the k = 3 resblock
    convs at C = 128.  Measured per layer at B = 32 (tools/probe_conv.py, profiles/archive/r02/r02y_probe_conv_b32.log): fused / pair =
    0.80-0.96 at C = 64, 0.86-0.91 at C = 32, 0.89-0.92 at C = 128 k = 3; 1.04-1.19 everywhere else.  The same rule lives in
    csrc/st2_engine.hip (conv())."""
    return pro != PRO_NONE and (C_in <= FUSED_MAX_C or (ks <= 3 and C_in <= FUSED_K3_MAX_C))


def conv_path():
    """How convs with a prologue over split-f16 weights are issued: "xs" = st2_act_split + st2_conv1d_xs (activation in an
    HBM-bound pass of its own, pure MFMA conv, InstanceNorm partial sums from the conv epilogue) unless `prefer_fused`
    routes the layer to st2_conv1d_f16s; "fused" (contract tests only, _hooks.py) = st2_conv1d_f16s everywhere."""
    return _hooks.conv_path


class XsTensor:
    """Pre-activated, pre-split conv operand written by `activate`: `data` is float16 [B, 2, cg, Lp, 8]
    (plane 0 = hi, 1 = lo; 16-byte slots of 8 channels), logical shape [B, C, L], `halo` zero columns in front."""

    def __init__(self, data, C, L, halo, x_scale=F16S_X_SCALE):
        self.data, self.C, self.L, self.halo, self.x_scale = data, C, L, halo, x_scale

    @property
    def cg(self):
        return self.data.shape[2]

    @property
    def Lp(self):
        return self.data.shape[3]


def xs_row_slots(L):
    """Slots per xs row for a tensor of length L: halo + L rounded up to the widest conv tile (512) + room for the
    last tile's taps (checked again inside st2_conv1d_xs)."""
    return XS_HALO + (max(L, 1) + 1 + 511) // 512 * 512 + 96


def activate(x, *, pro=PRO_NONE, slope=0.0, stats=None, gamma=None, beta=None, gamma_plus_one=False, alpha=None,
              c_pad=32, gb_seg=0, x_scale=None):
    """`st2_act_split`: x [B, C, L] fp32 -> XsTensor holding split_f16(x_scale * pro(x)) with the conv's zero padding
    (x_scale = x_scale_for(pro) unless the caller passes a calibrated power of two, `calibrated_x_scale`).  `gb_seg` > 0 (PRO_COLNORM on a token-merged view [1, C, G * gb_seg]): gamma / beta
    are [G, C] and row l // gb_seg applies at position l (per-utterance AdaLayerNorm affine, include/st2.h)."""
    lib = _lib.load()
    _chk(x, "x", 3)
    B, Cc, L = x.shape
    cg = (Cc + c_pad - 1) // c_pad * c_pad // 8
    Lp = xs_row_slots(L)
    data = torch.empty((B, 2, cg, Lp, 8), device=x.device, dtype=torch.float16)
    gbs = 0
    if pro in (PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE, PRO_COLNORM):
        _chk(stats, "stats", 3)
        _chk(gamma, "gamma", 2)
        _chk(beta, "beta", 2)
        want = (B, L, 2) if pro == PRO_COLNORM else (B, Cc, 2)
        assert tuple(stats.shape) == want and stats.is_contiguous(), (stats.shape, want)
        assert gamma.shape[1] == Cc and beta.shape[1] == Cc
        gbs = gamma.stride(0) if gamma.shape[0] > 1 else 0
        bbs = beta.stride(0) if beta.shape[0] > 1 else 0
        if gb_seg:
            assert pro == PRO_COLNORM and B == 1 and gamma.shape[0] * gb_seg >= L and beta.shape[0] == gamma.shape[0]
            assert gbs == bbs
        else:
            assert gamma.shape[0] in (1, B) and beta.shape[0] == gamma.shape[0] and gbs == bbs
    if pro in (PRO_ADAIN_SNAKE, PRO_SNAKE):
        _chk(alpha, "alpha", 1)
        assert alpha.numel() == Cc and alpha.is_contiguous()
    xsc = float(x_scale) if x_scale else x_scale_for(pro)
    _lib.check(lib.st2_act_split(x.data_ptr(), x.stride(0), x.stride(1), B, Cc, L, pro, slope, _ptr(stats),
                                 _ptr(gamma), _ptr(beta), gbs, int(gb_seg), 1 if gamma_plus_one else 0, _ptr(alpha),
                                 xsc, data.data_ptr(), cg, Lp, XS_HALO, _stream()), "st2_act_split")
    return XsTensor(data, Cc, L, XS_HALO, xsc)


def new_part(B, C, nt, device):
    """Buffer for a producer's per-slot partial sums: float2 [B * C][nt] (sum, sum of squares of y - shift) followed by float
    [B * C][nt] (the shifts = each slot's first stored value), include/st2.h `d.part`."""
    return torch.empty((B * C * nt * 3,), device=device, dtype=torch.float32)


def stats_finalize(part, B, C, nt, L, cols=128, eps=1e-5, out=None):
    """`st2_stats_finalize`: the buffer of `new_part` filled by the producer of a [B, C, L] tensor (slots of `cols` columns) ->
    stats [B, C, 2] (mean, rstd) of that tensor."""
    lib = _lib.load()
    assert part.numel() >= B * C * nt * 3 and part.is_contiguous()
    if out is None:
        out = torch.empty((B, C, 2), device=part.device, dtype=torch.float32)
    _lib.check(lib.st2_stats_finalize(part.data_ptr(), B * C, nt, L, eps, out.data_ptr(), int(cols), _stream()), "st2_stats_finalize")
    return out


def conv1d_xs(xs, wt, C_out, ks, *, dil=1, pad_left=0, L_out=None, bias=None, out=None, res=None, res_shift=0,
              res2=None, div=1.0, act=ACT_NONE, act_split=0, act_slope=0.0, want_stats=False, part_cols=None):
    """`st2_conv1d_xs` on an XsTensor; with want_stats returns (out, stats [B, C_out, 2]) where the InstanceNorm
    statistics of `out` come from the conv epilogue's per-tile partial sums + `st2_stats_finalize`."""
    lib = _lib.load()
    assert isinstance(xs, XsTensor) and isinstance(wt, SplitConvWeight)
    B, C_in, L_in = xs.data.shape[0], xs.C, xs.L
    if (wt.C_in, wt.C_out, wt.ks) != (C_in, C_out, ks) or not wt.wq.is_cuda or not wt.wq.is_contiguous():
        raise _lib.St2Error("split weight is for (C_in=%d, C_out=%d, ks=%d) on %s, call has (%d, %d, %d)" % (
            wt.C_in, wt.C_out, wt.ks, wt.wq.device, C_in, C_out, ks))
    if L_out is None:
        L_out = L_in
    if out is None:
        out = torch.empty((B, C_out, L_out), device=xs.data.device, dtype=torch.float32)
    _chk(out, "out", 3)
    assert out.shape == (B, C_out, L_out), (out.shape, (B, C_out, L_out))
    d = ConvDesc()
    d.B, d.C_in, d.C_out, d.L_in, d.L_out, d.ks, d.dil, d.pad_left = B, C_in, C_out, L_in, L_out, ks, dil, pad_left
    d.xs, d.xs_cg, d.xs_lp, d.xs_halo = xs.data.data_ptr(), xs.cg, xs.Lp, xs.halo
    d.wq, d.wq_co_pad, d.wq_cin_pad = wt.wq.data_ptr(), wt.co_pad, wt.cin_pad
    d.x_scale, d.out_scale, d.w_row_scale = xs.x_scale, 1.0 / xs.x_scale, wt.row_scale.data_ptr()
    _chk(bias, "bias", 1)
    d.bias = _ptr(bias)
    d.y, d.y_bs, d.y_cs = out.data_ptr(), out.stride(0), out.stride(1)
    _fill_epilogue(d, B, C_out, L_out, res, res_shift, res2, div, act, act_split, act_slope)
    part = None
    if want_stats:
        # 128, or 64 / 32 on a small grid (same rule as the C++ plans: bitwise); `part_cols=128` (tests) keeps the 128-column tiles
        pc = part_cols or lib.st2_conv1d_xs_part_cols(C.byref(d))
        nt = (L_out + pc - 1) // pc
        part = new_part(B, C_out, nt, out.device)
        d.part, d.part_nt, d.part_cols = part.data_ptr(), nt, pc
    _launch_conv(lib.st2_conv1d_xs, "st2_conv1d_xs", d)
    if want_stats:
        return out, stats_finalize(part, B, C_out, nt, L_out, cols=d.part_cols or 128)
    return out


def _fill_epilogue(d, B, C_out, L_out, res, res_shift, res2, div, act, act_split, act_slope):
    if res is not None:
        _chk(res, "res", 3)
        assert res.shape[0] == B and res.shape[1] == C_out and res.shape[2] == (L_out + (1 << res_shift) - 1 >> res_shift)
        d.res, d.res_bs, d.res_cs, d.res_shift = res.data_ptr(), res.stride(0), res.stride(1), res_shift
    if res2 is not None:
        _chk(res2, "res2", 3)
        assert res2.shape == (B, C_out, L_out)
        d.res2, d.res2_bs, d.res2_cs = res2.data_ptr(), res2.stride(0), res2.stride(1)
    d.div = div
    d.act, d.act_split, d.act_slope = act, act_split, act_slope


def _launch_conv(fn, fname, d):
    _lib.check(fn(C.byref(d), _stream()), fname)


def conv1d(x, wt, C_out, ks, *, dil=1, pad_left=0, L_out=None, bias=None, out=None,
           pro=PRO_NONE, slope=0.0, stats=None, gamma=None, beta=None, gamma_plus_one=False, alpha=None,
           res=None, res_shift=0, res2=None, div=1.0, act=ACT_NONE, act_split=0, act_slope=0.0, want_stats=False,
           gb_seg=0, x_scale=None):
    """Fused Conv1d, see `st2_conv1d` / `st2_conv1d_f16s` / `st2_conv1d_xs` in include/st2.h.  wt is either the
    packed K-major fp32 weight [C_in*ks, w_ld] of weights.pack_conv() (exact-fp32 MFMA kernel) or a
    weights.SplitConvWeight from weights.pack_conv_f16s() (split-f16 MFMA kernels, fp32-class accuracy at 5.3x the
    rate ceiling).  With a SplitConvWeight, a prologue and conv_path() == "xs" the call is issued as
    st2_act_split + st2_conv1d_xs.  want_stats=True returns (out, InstanceNorm statistics of out [B, C_out, 2])."""
    lib = _lib.load()
    _chk(x, "x", 3)
    B, C_in, L_in = x.shape
    split = isinstance(wt, SplitConvWeight)
    if (split and pad_left <= XS_HALO and L_in >= XS_MIN_L and (pro != PRO_NONE or C_in >= XS_MIN_C_PLAIN)
            and conv_path() == "xs" and not prefer_fused(pro, C_in, ks)):
        xs = activate(x, pro=pro, slope=slope, stats=stats, gamma=gamma, beta=beta,
                      gamma_plus_one=gamma_plus_one, alpha=alpha, gb_seg=gb_seg, x_scale=x_scale)
        return conv1d_xs(xs, wt, C_out, ks, dil=dil, pad_left=pad_left, L_out=L_out, bias=bias, out=out, res=res,
                         res_shift=res_shift, res2=res2, div=div, act=act, act_split=act_split, act_slope=act_slope,
                         want_stats=want_stats)
    if gb_seg:
        raise _lib.St2Error("gb_seg (per-segment affine of a token-merged view) exists on the st2_act_split + "
                            "st2_conv1d_xs path only; this call routes to the fused kernel")
    if split:
        if (wt.C_in, wt.C_out, wt.ks) != (C_in, C_out, ks) or not wt.wq.is_cuda or not wt.wq.is_contiguous():
            raise _lib.St2Error("split weight is for (C_in=%d, C_out=%d, ks=%d) on %s, call has (%d, %d, %d)" % (
                wt.C_in, wt.C_out, wt.ks, wt.wq.device, C_in, C_out, ks))
    else:
        _chk(wt, "wt", 2)
        if wt.shape[0] != C_in * ks or not wt.is_contiguous():
            raise _lib.St2Error("packed weight has shape %s, expected [%d, >=%d]" % (tuple(wt.shape), C_in * ks,
                                                                                      C_out))
    if L_out is None:
        L_out = L_in
    if out is None:
        out = torch.empty((B, C_out, L_out), device=x.device, dtype=torch.float32)
    _chk(out, "out", 3)
    assert out.shape == (B, C_out, L_out), (out.shape, (B, C_out, L_out))
    d = ConvDesc()
    d.B, d.C_in, d.C_out, d.L_in, d.L_out, d.ks, d.dil, d.pad_left = B, C_in, C_out, L_in, L_out, ks, dil, pad_left
    d.x, d.x_bs, d.x_cs = x.data_ptr(), x.stride(0), x.stride(1)
    if split:
        d.wq, d.wq_co_pad, d.wq_cin_pad = wt.wq.data_ptr(), wt.co_pad, wt.cin_pad
        d.x_scale = float(x_scale) if x_scale else x_scale_for(pro)
        d.out_scale, d.w_row_scale = 1.0 / d.x_scale, wt.row_scale.data_ptr()
        fn, fname = lib.st2_conv1d_f16s, "st2_conv1d_f16s"
    else:
        d.wt, d.w_ld = wt.data_ptr(), wt.shape[1]
        fn, fname = lib.st2_conv1d, "st2_conv1d"
    _chk(bias, "bias", 1)
    d.bias = _ptr(bias)
    d.y, d.y_bs, d.y_cs = out.data_ptr(), out.stride(0), out.stride(1)
    d.pro, d.slope = pro, slope
    if pro in (PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE, PRO_COLNORM):
        _chk(stats, "stats", 3)
        _chk(gamma, "gamma", 2)
        _chk(beta, "beta", 2)
        want = (B, L_in, 2) if pro == PRO_COLNORM else (B, C_in, 2)
        assert tuple(stats.shape) == want and stats.is_contiguous(), (stats.shape, want)
        assert gamma.shape[1] == C_in and beta.shape[1] == C_in
        gbs = gamma.stride(0) if gamma.shape[0] > 1 else 0
        bbs = beta.stride(0) if beta.shape[0] > 1 else 0
        assert gamma.shape[0] in (1, B) and beta.shape[0] == gamma.shape[0] and (gbs == bbs)
        d.stats, d.gamma, d.beta, d.gb_bs = stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), gbs
        d.gamma_plus_one = 1 if gamma_plus_one else 0
    if pro in (PRO_ADAIN_SNAKE, PRO_SNAKE):
        _chk(alpha, "alpha", 1)
        assert alpha.numel() == C_in and alpha.is_contiguous()
        d.alpha = alpha.data_ptr()
    _fill_epilogue(d, B, C_out, L_out, res, res_shift, res2, div, act, act_split, act_slope)
    ws = part = None
    if split and want_stats:  # InstanceNorm statistics of the output from the epilogue's per-tile partial sums
        nt = (L_out + 127) // 128
        part = new_part(B, C_out, nt, out.device)
        d.part, d.part_nt = part.data_ptr(), nt
    elif split:  # skinny layers (few workgroups, long k loop) run split-K: the library says how much workspace it wants
        nb = lib.st2_conv1d_f16s_splitk_bytes(C.byref(d))
        if nb > 0:
            ws = torch.empty((nb,), device=x.device, dtype=torch.uint8)  # caching allocator: stream- and capture-safe
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), nb
    _launch_conv(fn, fname, d)
    if want_stats:
        return out, (stats_finalize(part, B, C_out, nt, L_out) if part is not None else instnorm_stats(out))
    return out


def conv1d_direct(x, w, bias, stride, pad, L_out=None, out=None):
    """Plain-weight ([C_out, C_in, ks]) direct conv for strided / tiny-C_in layers (`st2_conv1d_direct`)."""
    lib = _lib.load()
    _chk(x, "x", 3)
    _chk(w, "w", 3)
    _chk(bias, "bias", 1)
    assert w.is_contiguous()
    B, C_in, L_in = x.shape
    C_out, C_in_w, ks = w.shape
    assert C_in_w == C_in
    if L_out is None:
        L_out = (L_in + 2 * pad - ks) // stride + 1
    if out is None:
        out = torch.empty((B, C_out, L_out), device=x.device, dtype=torch.float32)
    _chk(out, "out", 3)
    _lib.check(lib.st2_conv1d_direct(x.data_ptr(), x.stride(0), x.stride(1), w.data_ptr(), _ptr(bias),
                                     out.data_ptr(), out.stride(0), out.stride(1), B, C_in, C_out, L_in, L_out,
                                     ks, stride, pad, _stream()), "st2_conv1d_direct")
    return out


def phase_split(x, stride, pad, Lu):
    """x [B, C, L] -> xp [B, C*stride, Lu], xp[b, c*stride + r, u] = x[b, c, u*stride + r - pad] (`st2_phase_split`)."""
    lib = _lib.load()
    _chk(x, "x", 3)
    B, Cc, L = x.shape
    xp = torch.empty((B, Cc * stride, Lu), device=x.device, dtype=torch.float32)
    _lib.check(lib.st2_phase_split(x.data_ptr(), x.stride(0), x.stride(1), B, Cc, L, stride, pad, xp.data_ptr(),
                                   xp.stride(0), xp.stride(1), Lu, _stream()), "st2_phase_split")
    return xp


def instnorm_stats(x, eps=1e-5, out=None):
    lib = _lib.load()
    _chk(x, "x", 3)
    B, Cc, L = x.shape
    if out is None:
        out = torch.empty((B, Cc, 2), device=x.device, dtype=torch.float32)
    _lib.check(lib.st2_instnorm_stats(x.data_ptr(), x.stride(0), x.stride(1), B, Cc, L, eps, out.data_ptr(),
                                      _stream()), "st2_instnorm_stats")
    return out


def colnorm_stats(x, eps=1e-5, out=None):
    lib = _lib.load()
    _chk(x, "x", 3)
    B, Cc, L = x.shape
    if out is None:
        out = torch.empty((B, L, 2), device=x.device, dtype=torch.float32)
    _lib.check(lib.st2_colnorm_stats(x.data_ptr(), x.stride(0), x.stride(1), B, Cc, L, eps, out.data_ptr(),
                                     _stream()), "st2_colnorm_stats")
    return out


def style_fc(s, wt, bias, act=ACT_NONE, out=None):
    """h = act(s @ wt + bias); wt is [K, J] (already transposed at pack time)."""
    lib = _lib.load()
    _chk(s, "s", 2)
    _chk(wt, "wt", 2)
    _chk(bias, "bias", 1)
    assert s.is_contiguous() and wt.is_contiguous()
    B, K = s.shape
    assert wt.shape[0] == K
    J = wt.shape[1]
    if out is None:
        out = torch.empty((B, J), device=s.device, dtype=torch.float32)
    _lib.check(lib.st2_style_fc(s.data_ptr(), B, K, wt.data_ptr(), _ptr(bias), J, act, out.data_ptr(), _stream()),
               "st2_style_fc")
    return out


CVT_TILE = 1024  # positions per st2_convt_interleave tile (= per entry of its partial-sum output)


def convt_interleave(phases, C_out, stride, pad, L_raw, bias=None, add=None, reflect_left=False, out=None,
                     want_stats=False):
    """`st2_convt_interleave[_stats]`; with want_stats returns (out, InstanceNorm statistics of out [B, C_out, 2]) from
    the kernel's per-tile partial sums + `st2_stats_finalize`."""
    lib = _lib.load()
    _chk(phases, "phases", 3)
    _chk(bias, "bias", 1)
    _chk(add, "add", 3)
    B, RC, Lq = phases.shape
    assert RC == stride * C_out
    L_out = L_raw + (1 if reflect_left else 0)
    if out is None:
        out = torch.empty((B, C_out, L_out), device=phases.device, dtype=torch.float32)
    if add is not None:
        assert add.shape == (B, C_out, L_out), (add.shape, (B, C_out, L_out))
    a_bs, a_cs = (add.stride(0), add.stride(1)) if add is not None else (0, 0)
    part, nt = None, 0
    if want_stats:
        nt = (L_out + CVT_TILE - 1) // CVT_TILE
        part = new_part(B, C_out, nt, phases.device)
    _lib.check(lib.st2_convt_interleave_stats(phases.data_ptr(), phases.stride(0), phases.stride(1), Lq, _ptr(bias),
                                              _ptr(add), a_bs, a_cs, out.data_ptr(), out.stride(0), out.stride(1), B,
                                              C_out, stride, pad, L_raw, 1 if reflect_left else 0, _ptr(part), nt,
                                              _stream()), "st2_convt_interleave")
    if want_stats:
        return out, stats_finalize(part, B, C_out, nt, L_out, cols=CVT_TILE)
    return out


def adain_leaky_pool(x, stats, gamma, beta, slope, w, bias, out=None):
    lib = _lib.load()
    _chk(x, "x", 3)
    _chk(stats, "stats", 3)
    _chk(gamma, "gamma", 2)
    _chk(beta, "beta", 2)
    _chk(w, "w", 2)
    _chk(bias, "bias", 1)
    B, Cc, L = x.shape
    assert w.shape == (Cc, 3) and w.is_contiguous() and gamma.stride(0) == beta.stride(0)
    if out is None:
        out = torch.empty((B, Cc, 2 * L), device=x.device, dtype=torch.float32)
    _lib.check(lib.st2_adain_leaky_pool(x.data_ptr(), x.stride(0), x.stride(1), stats.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), gamma.stride(0), slope, w.data_ptr(), _ptr(bias),
                                        out.data_ptr(), out.stride(0), out.stride(1), B, Cc, L, _stream()),
               "st2_adain_leaky_pool")
    return out


def har_source(f0, U, noise, lin_w, lin_b, sine_amp=0.1, noise_std=0.003, voiced_threshold=10.0,
               sample_rate=24000.0):
    """f0 [B, F] -> har_source [B, F*U]; noise [B, F*U, H] standard-normal draws."""
    lib = _lib.load()
    _chk(f0, "f0", 2)
    _chk(noise, "noise", 3)
    _chk(lin_w, "lin_w")
    _chk(lin_b, "lin_b")
    B, F = f0.shape
    H = noise.shape[2]
    assert f0.is_contiguous() and noise.is_contiguous() and noise.shape == (B, F * U, H)
    assert lin_w.numel() == H and lin_w.is_contiguous()
    scratch = torch.empty((B, H, F), device=f0.device, dtype=torch.float32)
    out = torch.empty((B, F * U), device=f0.device, dtype=torch.float32)
    _lib.check(lib.st2_har_source(f0.data_ptr(), B, F, U, H, noise.data_ptr(), lin_w.data_ptr(), lin_b.data_ptr(),
                                  sine_amp, noise_std, voiced_threshold, sample_rate, scratch.data_ptr(),
                                  out.data_ptr(), _stream()), "st2_har_source")
    return out


def stft_mag_phase(x, n_fft, hop):
    lib = _lib.load()
    _chk(x, "x", 2)
    assert x.is_contiguous()
    B, L = x.shape
    M = L // hop + 1
    har = torch.empty((B, n_fft + 2, M), device=x.device, dtype=torch.float32)
    _lib.check(lib.st2_stft_mag_phase(x.data_ptr(), B, L, n_fft, hop, har.data_ptr(), har.stride(0), har.stride(1),
                                      _stream()), "st2_stft_mag_phase")
    return har


def istft(sp, n_fft, hop):
    """sp [B, n_fft+2, M] = cat(spec, phase) -> wave [B, 1, hop*(M-1)]."""
    lib = _lib.load()
    _chk(sp, "sp", 3)
    B, Cc, M = sp.shape
    assert Cc == n_fft + 2
    wave = torch.empty((B, 1, hop * (M - 1)), device=sp.device, dtype=torch.float32)
    _lib.check(lib.st2_istft(sp.data_ptr(), sp.stride(0), sp.stride(1), B, M, n_fft, hop, wave.data_ptr(),
                             wave.stride(0), _stream()), "st2_istft")
    return wave


def attention(q, k, v, heads, scale, out=None, key_len=None):
    """q, k, v: [B, heads*D, N] views with identical strides -> [B, heads*D, N].  key_len (int32 [B] on the device):
    keys m >= key_len[b] are padding and excluded from the softmax (`st2_attention_keylen`)."""
    lib = _lib.load()
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, n, 3)
    B, HD, N = q.shape
    D = HD // heads
    assert k.shape == q.shape and v.shape == q.shape
    assert _bs_cs(q) == _bs_cs(k) == _bs_cs(v)
    if out is None:
        out = torch.empty((B, HD, N), device=q.device, dtype=torch.float32)
    if key_len is not None:
        assert key_len.is_cuda and key_len.dtype == torch.int32 and key_len.numel() == B and key_len.is_contiguous()
    _lib.check(lib.st2_attention_keylen(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(1),
                                        out.data_ptr(), out.stride(0), out.stride(1), B, heads, D, N, scale,
                                        0 if key_len is None else key_len.data_ptr(), _stream()), "st2_attention")
    return out


def colnorm_apply(x, stats, gamma, beta, *, gamma_plus_one=False, act=ACT_NONE, slope=0.0, lengths=None, out=None):
    """LayerNorm over channels applied (`st2_colnorm_apply`): x [B, C, L], stats [B, L, 2] from colnorm_stats,
    gamma / beta [1 or B, C]; positions l >= lengths[b] (int32 [B] on the device) are written as zero."""
    lib = _lib.load()
    _chk(x, "x", 3)
    _chk(stats, "stats", 3)
    _chk(gamma, "gamma", 2)
    _chk(beta, "beta", 2)
    B, Cc, L = x.shape
    assert tuple(stats.shape) == (B, L, 2) and stats.is_contiguous()
    assert gamma.shape[1] == Cc and beta.shape == gamma.shape and gamma.shape[0] in (1, B)
    gbs = gamma.stride(0) if gamma.shape[0] > 1 else 0
    assert (beta.stride(0) if beta.shape[0] > 1 else 0) == gbs and gamma.stride(1) == 1 and beta.stride(1) == 1
    if lengths is not None:
        assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.numel() == B and lengths.is_contiguous()
    if out is None:
        out = torch.empty((B, Cc, L), device=x.device, dtype=torch.float32)
    _chk(out, "out", 3)
    _lib.check(lib.st2_colnorm_apply(x.data_ptr(), x.stride(0), x.stride(1), stats.data_ptr(), gamma.data_ptr(),
                                     beta.data_ptr(), gbs, 1 if gamma_plus_one else 0, act, slope,
                                     0 if lengths is None else lengths.data_ptr(), out.data_ptr(), out.stride(0),
                                     out.stride(1), B, Cc, L, _stream()), "st2_colnorm_apply")
    return out


_last_lstm_scratch = None  # scratch of the most recent cooperative launch (tests read its status word)


def lstm_mode():
    """"coop": W_hh register-resident over 8 CUs per group, one hidden-state exchange per step (st2_lstm_bidir_coop);
    "single" (tests only, _hooks.py): one CU per (utterance, direction) streaming W_hh from L2 (st2_lstm_bidir) -- the
    kernel the library itself falls back to when a device cannot hold a cooperative launch co-resident."""
    return _hooks.lstm


def lstm_bidir(G, whh_t, lengths=None, out=None):
    """G [B, 8H, N] projected inputs (both directions) -> Y [B, 2H, N]; lengths: int32 [B] on the device or None.
    The cooperative kernel is used when the library accepts the launch (its workgroups must all be co-resident: the
    library checks the device's occupancy and refuses otherwise -- then, and for B > 48, the single-CU kernel runs).
    A cooperative group that times out is repaired in-stream (st2_lstm_bidir_coop_recovering: the single-CU kernel re-runs
    the call into the same output; STATUS_LSTM_RECOVERED, a warning from `check_status()`)."""
    global _last_lstm_scratch
    lib = _lib.load()
    _chk(G, "G", 3)
    _chk(whh_t, "whh_t", 3)
    B, R, N = G.shape
    H = R // 8
    assert whh_t.shape == (2, H, 4 * H) and whh_t.is_contiguous()
    if lengths is not None:
        assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.numel() == B and lengths.is_contiguous()
    if out is None:
        out = torch.empty((B, 2 * H, N), device=G.device, dtype=torch.float32)
    lp = 0 if lengths is None else lengths.data_ptr()
    nbytes = lib.st2_lstm_coop_scratch_bytes(B) if lstm_mode() == "coop" else 0
    if nbytes > 0:
        scratch = torch.empty((nbytes,), device=G.device, dtype=torch.uint8)
        fn = lib.st2_lstm_bidir_coop_recovering if _hooks.lstm_recover else lib.st2_lstm_bidir_coop
        rc = fn(G.data_ptr(), G.stride(0), G.stride(1), whh_t.data_ptr(), lp, B, H, N, out.data_ptr(), out.stride(0),
                out.stride(1), scratch.data_ptr(), nbytes, _stream())
        if rc == 0:
            _last_lstm_scratch = scratch
            return out
        msg = (lib.st2_last_error() or b"").decode()
        if "co-resident" not in msg:
            raise _lib.St2Error("st2_lstm_bidir_coop failed: %s" % msg)
        # refused (host-side occupancy check, nothing was launched): the single-CU kernel for THIS call only -- the
        # answer depends on the device and the batch, so it is asked again next time
    _lib.check(lib.st2_lstm_bidir(G.data_ptr(), G.stride(0), G.stride(1), whh_t.data_ptr(), lp, B, H, N,
                                  out.data_ptr(), out.stride(0), out.stride(1), _stream()), "st2_lstm_bidir")
    return out


def mfma_load(kind=0, workgroups=720, iters=400):
    """Queues one launch of the library's matrix-pipe load generator on the current stream (include/st2.h
    `st2_probe_mfma_stream`: the MFMA cadences next to which round 5's BiLSTM kernels returned wrong bits).  The co-residency
    canaries (tests/test_zz_coresidency_gpu.py, `pipeline.coresidency_selfcheck`) run product kernels on another stream
    meanwhile and demand the idle result bit for bit."""
    _lib.check(_lib.load().st2_probe_mfma_stream(int(kind), int(workgroups), int(iters), _stream()), "st2_probe_mfma_stream")


def lstm_coop_status():
    """Status word of the most recent cooperative LSTM launch (synchronises): 0 = ok, 1 = a spin timed out."""
    if _last_lstm_scratch is None:
        return 0
    return int(_last_lstm_scratch[:4].view(torch.int32).item())


def add_chanvec(x, v, out=None):
    lib = _lib.load()
    _chk(x, "x", 3)
    _chk(v, "v", 2)
    B, Cc, N = x.shape
    assert v.shape == (B, Cc)
    if out is None:
        out = torch.empty((B, Cc, N), device=x.device, dtype=torch.float32)
    _lib.check(lib.st2_add_chanvec(x.data_ptr(), x.stride(0), x.stride(1), v.data_ptr(), v.stride(0), out.data_ptr(),
                                   out.stride(0), out.stride(1), B, Cc, N, _stream()), "st2_add_chanvec")
    return out


def mean_tokens(x, out=None, lengths=None):
    """m[b, c] = mean over the first lengths[b] tokens (all N when lengths is None; int32 [B] on the device)."""
    lib = _lib.load()
    _chk(x, "x", 3)
    B, Cc, N = x.shape
    if out is None:
        out = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    if lengths is not None:
        assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.numel() == B and lengths.is_contiguous()
    _lib.check(lib.st2_mean_tokens_len(x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(), out.stride(0), B, Cc,
                                       N, 0 if lengths is None else lengths.data_ptr(), _stream()), "st2_mean_tokens")
    return out


def axpbypcz(x, a, y=None, b=0.0, z=None, c=0.0, out=None):
    """out = a*x + b*y + c*z (flat, contiguous)."""
    lib = _lib.load()
    _chk(x, "x")
    assert x.is_contiguous()
    for t in (y, z):
        if t is not None:
            _chk(t, "operand")
            assert t.is_contiguous() and t.numel() == x.numel()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.st2_axpbypcz(x.data_ptr(), a, _ptr(y), b, _ptr(z), c, out.data_ptr(), x.numel(), _stream()),
               "st2_axpbypcz")
    return out


def time_features(t, w, B, out=None):
    """`st2_time_features`: [B, 1 + 2*len(w)] = [t, sin(t w 2 pi), cos(t w 2 pi)] (denoiser time embedding input)."""
    lib = _lib.load()
    _chk(w, "w", 1)
    H2 = w.numel()
    if out is None:
        out = torch.empty((B, 1 + 2 * H2), device=w.device, dtype=torch.float32)
    _lib.check(lib.st2_time_features(float(t), w.data_ptr(), H2, B, out.data_ptr(), _stream()), "st2_time_features")
    return out


def tokens_to_channels(e, out, B=None):
    """`st2_tokens_to_channels`: e [B, N, E] (or [N, E] broadcast over `B`) -> out[b, :E, :N] = e[b].T, `out` an NCL view."""
    lib = _lib.load()
    _chk(e, "e")
    _chk(out, "out", 3)
    assert e.is_contiguous()
    if e.dim() == 2:
        N, E = e.shape
        e_bs = 0
        B = out.shape[0] if B is None else B
    else:
        B, N, E = e.shape
        e_bs = N * E
    assert out.shape[0] == B and out.shape[1] == E and out.shape[2] == N, (out.shape, (B, E, N))
    _lib.check(lib.st2_tokens_to_channels(e.data_ptr(), e_bs, B, N, E, out.data_ptr(), out.stride(0), out.stride(1),
                                          _stream()), "st2_tokens_to_channels")
    return out


def broadcast_cols(x, out):
    """`st2_broadcast_cols`: out[b, c, n] = x[b, c] for every n; x [B, C] (unit stride along C), out an NCL view."""
    lib = _lib.load()
    _chk(x, "x", 2)
    _chk(out, "out", 3)
    B, Cc = x.shape
    assert out.shape[0] == B and out.shape[1] == Cc
    _lib.check(lib.st2_broadcast_cols(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), out.stride(1), B, Cc,
                                      out.shape[2], _stream()), "st2_broadcast_cols")
    return out


def copy_ncl(x, out):
    """`st2_copy_ncl`: strided copy between NCL views of equal shape."""
    lib = _lib.load()
    _chk(x, "x", 3)
    _chk(out, "out", 3)
    assert x.shape == out.shape
    B, Cc, L = x.shape
    _lib.check(lib.st2_copy_ncl(x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(), out.stride(0), out.stride(1), B,
                                Cc, L, _stream()), "st2_copy_ncl")
    return out


def duration_head(x, w, bias, lengths=None, tail=0, want_sums=False):
    """`st2_duration_head`: x [B, K, N] channel-major, w [J, K], bias [J] -> int64 durations [B, N] (and the un-rounded
    sigmoid sums when want_sums)."""
    lib = _lib.load()
    _chk(x, "x", 3)
    _chk(w, "w", 2)
    _chk(bias, "bias", 1)
    B, K, N = x.shape
    J = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous() and bias.numel() == J
    if lengths is not None:
        assert lengths.is_cuda and lengths.dtype == torch.int32 and lengths.numel() == B and lengths.is_contiguous()
    dur = torch.empty((B, N), device=x.device, dtype=torch.int64)
    sums = torch.empty((B, N), device=x.device, dtype=torch.float32) if want_sums else None
    _lib.check(lib.st2_duration_head(x.data_ptr(), x.stride(0), x.stride(1), w.data_ptr(), bias.data_ptr(), B, K, J, N,
                                     0 if lengths is None else lengths.data_ptr(), int(tail), dur.data_ptr(),
                                     _ptr(sums), _stream()), "st2_duration_head")
    return (dur, sums) if want_sums else dur


def expand_by_durations(x, dur, T, shift=False, out=None):
    """`st2_expand_by_durations`: x [B, C, N], dur int64 [B, N] (rows sum to T) -> [B, C, T]."""
    lib = _lib.load()
    _chk(x, "x", 3)
    B, Cc, N = x.shape
    assert dur.is_cuda and dur.dtype == torch.int64 and dur.shape == (B, N) and dur.is_contiguous()
    if out is None:
        out = torch.empty((B, Cc, T), device=x.device, dtype=torch.float32)
    _chk(out, "out", 3)
    _lib.check(lib.st2_expand_by_durations(x.data_ptr(), x.stride(0), x.stride(1), dur.data_ptr(), B, Cc, N, T,
                                           1 if shift else 0, out.data_ptr(), out.stride(0), out.stride(1), _stream()),
               "st2_expand_by_durations")
    return out


# ---- reference-audio style path (st2_style.hip) ----------------------------------------------------------------------
def stft_frames(wave, n_win, hop, shift):
    """`st2_stft_frames`: wave [B, L] -> frames [B, n_win, L // hop + 1] (reflect-padded frame columns of torch.stft)."""
    lib = _lib.load()
    _chk(wave, "wave", 2)
    B, L = wave.shape
    M = L // hop + 1
    fr = torch.empty((B, n_win, M), device=wave.device, dtype=torch.float32)
    _lib.check(lib.st2_stft_frames(wave.data_ptr(), wave.stride(0), B, L, n_win, hop, shift, fr.data_ptr(), fr.stride(0),
                                   fr.stride(1), _stream()), "st2_stft_frames")
    return fr


def power_spectrum(y):
    """`st2_power_spectrum`: y [B, 2K, M] (real rows then imaginary rows) -> [B, K, M]."""
    lib = _lib.load()
    _chk(y, "y", 3)
    B, K2, M = y.shape
    assert K2 % 2 == 0
    p = torch.empty((B, K2 // 2, M), device=y.device, dtype=torch.float32)
    _lib.check(lib.st2_power_spectrum(y.data_ptr(), y.stride(0), y.stride(1), B, K2 // 2, M, p.data_ptr(), p.stride(0),
                                      p.stride(1), _stream()), "st2_power_spectrum")
    return p


def log_norm_(x, eps, mean, std):
    """`st2_log_norm`: x = (log(eps + x) - mean) / std in place (x contiguous)."""
    lib = _lib.load()
    _chk(x, "x")
    assert x.is_contiguous()
    _lib.check(lib.st2_log_norm(x.data_ptr(), x.numel(), eps, mean, std, _stream()), "st2_log_norm")
    return x


def _chk_map(t, name):
    _chk(t, name, 4)  # [B, H, C, W] view, W contiguous


def dwconv3x3s2(x, w, bias, out):
    """`st2_dwconv3x3s2`: x [B, H, C, W] (any strides, W contiguous), w [C, 3, 3], bias [C] -> out [B, Ho, C, Wo]."""
    lib = _lib.load()
    _chk_map(x, "x")
    _chk_map(out, "out")
    _chk(w, "w", 3)
    _chk(bias, "bias", 1)
    B, H, Cc, Wd = x.shape
    assert w.shape == (Cc, 3, 3) and w.is_contiguous()
    assert out.shape == (B, (H - 1) // 2 + 1, Cc, (Wd - 1) // 2 + 1), (out.shape, x.shape)
    _lib.check(lib.st2_dwconv3x3s2(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), w.data_ptr(), _ptr(bias), B, Cc,
                                   H, Wd, out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), _stream()),
               "st2_dwconv3x3s2")
    return out


def avgpool2x2(x, out):
    """`st2_avgpool2x2`: x [B, H, C, W] -> out [B, H/2, C, (W+1)/2] (odd widths replicate their last column)."""
    lib = _lib.load()
    _chk_map(x, "x")
    _chk_map(out, "out")
    B, H, Cc, Wd = x.shape
    assert out.shape == (B, H // 2, Cc, (Wd + 1) // 2), (out.shape, x.shape)
    _lib.check(lib.st2_avgpool2x2(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), B, Cc, H, Wd, out.data_ptr(),
                                  out.stride(0), out.stride(1), out.stride(2), _stream()), "st2_avgpool2x2")
    return out
