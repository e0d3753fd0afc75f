"""Per-kernel parity: every C-ABI entry point (through styletts2_amd.ops) against its fp32
PyTorch-CPU contract in oracle/ops_ref.py, on seeded random data.  Tolerances are written per test;
they are fp32 round-off class (different summation order), not a precision downgrade."""
import math

import pytest
import torch

from oracle import ops_ref as R
from styletts2_amd import _hooks, ops, weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(t):
    return None if t is None else t.to(DEV)


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def make_conv_case(seed, B, C_in, C_out, L, ks, dil, pro, act=R.ACT_NONE, res=False, res2=False, res_shift=0,
                   div=1.0, pad_left=None, L_out=None, bias=True, sliced=False):
    gen = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=gen)
    x = r(B, C_in, L) * 1.5 + 0.3
    w = r(C_out, C_in, ks) / math.sqrt(C_in * ks)
    kw = dict(dil=dil, pad_left=(ks - 1) * dil // 2 if pad_left is None else pad_left, L_out=L_out,
              bias=r(C_out) if bias else None, pro=pro, div=div, act=act)
    Lo = L if L_out is None else L_out
    if pro in (R.PRO_LEAKY, R.PRO_ADAIN_LEAKY):
        kw["slope"] = 0.2
    if pro in (R.PRO_ADAIN_LEAKY, R.PRO_ADAIN_SNAKE):
        kw["stats"] = R.instnorm_stats(x)
        h = r(B, 2 * C_in + 3) * 0.5
        kw["gamma"], kw["beta"] = h[:, 1:1 + C_in], h[:, 1 + C_in:1 + 2 * C_in]
    if pro == R.PRO_COLNORM:
        kw["stats"] = R.colnorm_stats(x)
        kw["gamma"], kw["beta"] = r(1, C_in), r(1, C_in)
    if pro in (R.PRO_ADAIN_SNAKE, R.PRO_SNAKE):
        kw["alpha"] = torch.rand(C_in, generator=gen) + 0.5
    if res:
        kw["res"] = r(B, C_out, (Lo + (1 << res_shift) - 1) >> res_shift)
        kw["res_shift"] = res_shift
    if res2:
        kw["res2"] = r(B, C_out, Lo)
    if act == R.ACT_EXP_SIN:
        kw["act_split"] = C_out // 2
    if act == R.ACT_LEAKY:
        kw["act_slope"] = 0.1
    return x, w, kw


CONV_CASES = [
    # B, C_in, C_out, L, ks, dil, pro, extra
    dict(B=2, C_in=8, C_out=32, L=64, ks=3, dil=1, pro=R.PRO_NONE),
    dict(B=1, C_in=128, C_out=128, L=300, ks=3, dil=1, pro=R.PRO_ADAIN_SNAKE, res=True),
    dict(B=2, C_in=128, C_out=128, L=1001, ks=7, dil=3, pro=R.PRO_ADAIN_SNAKE, res=True, res2=True, div=3.0),
    dict(B=1, C_in=128, C_out=128, L=777, ks=11, dil=5, pro=R.PRO_ADAIN_SNAKE),
    dict(B=1, C_in=256, C_out=256, L=500, ks=11, dil=1, pro=R.PRO_ADAIN_SNAKE, res=True),
    dict(B=2, C_in=66, C_out=40, L=130, ks=3, dil=1, pro=R.PRO_ADAIN_LEAKY, res=True, res_shift=1,
         div=math.sqrt(2)),
    dict(B=1, C_in=1090, C_out=1024, L=100, ks=3, dil=1, pro=R.PRO_ADAIN_LEAKY),
    dict(B=2, C_in=130, C_out=70, L=90, ks=1, dil=1, pro=R.PRO_NONE, bias=False),
    dict(B=2, C_in=1024, C_out=512, L=100, ks=1, dil=1, pro=R.PRO_COLNORM, act=R.ACT_GELU),
    dict(B=1, C_in=128, C_out=22, L=481, ks=7, dil=1, pro=R.PRO_LEAKY, act=R.ACT_EXP_SIN),
    dict(B=1, C_in=32, C_out=1, L=300, ks=7, dil=1, pro=R.PRO_SNAKE, act=R.ACT_TANH),
    dict(B=2, C_in=64, C_out=60, L=50, ks=2, dil=1, pro=R.PRO_LEAKY, pad_left=1, L_out=51),
    dict(B=1, C_in=512, C_out=512, L=37, ks=5, dil=1, pro=R.PRO_NONE, act=R.ACT_LEAKY),
    dict(B=3, C_in=2, C_out=3, L=5, ks=3, dil=1, pro=R.PRO_NONE),
    dict(B=1, C_in=768, C_out=2048, L=700, ks=1, dil=1, pro=R.PRO_NONE, act=R.ACT_GELU_TANH),
]


F16S_EXTRA_CASES = [
    # the three wave layouts (C_out > 64, 33..64, <= 32) x long tiles, channel tails, every kernel size
    dict(B=2, C_in=128, C_out=128, L=2500, ks=11, dil=5, pro=R.PRO_ADAIN_SNAKE, res=True, res2=True, div=3.0),
    dict(B=2, C_in=64, C_out=64, L=1300, ks=7, dil=3, pro=R.PRO_ADAIN_SNAKE, res=True),
    dict(B=2, C_in=32, C_out=32, L=2100, ks=3, dil=5, pro=R.PRO_ADAIN_SNAKE, res=True),
    dict(B=1, C_in=32, C_out=32, L=1100, ks=11, dil=1, pro=R.PRO_ADAIN_SNAKE),
    dict(B=1, C_in=514, C_out=1024, L=75, ks=3, dil=1, pro=R.PRO_ADAIN_LEAKY),
    dict(B=1, C_in=256, C_out=1280, L=161, ks=2, dil=1, pro=R.PRO_LEAKY, pad_left=1, L_out=162),
    dict(B=2, C_in=64, C_out=1, L=700, ks=7, dil=1, pro=R.PRO_SNAKE, act=R.ACT_TANH),
    dict(B=1, C_in=17, C_out=33, L=129, ks=5, dil=2, pro=R.PRO_NONE),
    # split-K launches of st2_conv1d_f16s (few workgroups, long k loop): every epilogue term through the reduction kernel
    dict(B=3, C_in=2048, C_out=1024, L=100, ks=1, dil=1, pro=R.PRO_NONE, res=True, res2=True, div=2.0),
    dict(B=1, C_in=1000, C_out=300, L=112, ks=1, dil=1, pro=R.PRO_COLNORM, res=True, act=R.ACT_GELU),
    dict(B=1, C_in=1024, C_out=2048, L=87, ks=1, dil=1, pro=R.PRO_COLNORM, act=R.ACT_GELU_TANH),
    # ... the slices are stored in the accumulator layout of the tile: all three wave layouts, several column / row tiles, ragged ends
    dict(B=2, C_in=512, C_out=300, L=301, ks=1, dil=1, pro=R.PRO_NONE, res=True, act=R.ACT_GELU),       # 128 x 128 tiles, 3 x 3 of them
    dict(B=1, C_in=512, C_out=40, L=601, ks=1, dil=1, pro=R.PRO_LEAKY, res=True, res2=True, div=2.0),  # 64 x 256 tiles (2 x 2 waves)
    dict(B=1, C_in=520, C_out=24, L=1100, ks=3, dil=2, pro=R.PRO_LEAKY, res=True),                    # 32 x 512 tiles (1 x 4 waves)
]


@pytest.mark.parametrize("kernel", ["f32", "f16s", "xs"])
@pytest.mark.parametrize("case", CONV_CASES + F16S_EXTRA_CASES, ids=lambda c: "ci%d_co%d_L%d_k%d_d%d_p%d" % (
    c["C_in"], c["C_out"], c["L"], c["ks"], c["dil"], c["pro"]))
def test_conv1d_matches_contract(case, kernel, monkeypatch):
    """All conv kernels against the same contract: `f32` = exact-fp32 MFMA (st2_conv1d), `f16s` = split-f16 MFMA with
    the prologue fused (st2_conv1d_f16s), `xs` = activation pass + pure split-f16 MFMA conv (st2_act_split +
    st2_conv1d_xs).  The split contract carries the operand split (hi + lo of v * scale), so the bar is the same
    fp32 round-off class for all three."""
    monkeypatch.setattr(_hooks, "conv_path", "xs" if kernel == "xs" else "fused")
    x, w, kw = make_conv_case(seed=1234, **case)
    wt = weights.pack_conv(w) if kernel == "f32" else weights.pack_conv_f16s(w)
    C_out, ks = w.shape[0], w.shape[2]
    ref = R.conv1d(x, wt, C_out, ks, **kw)
    exact = R.conv1d(x.double(), weights.pack_conv(w).double(), C_out, ks,
                     **{k: (v.double() if torch.is_tensor(v) else v) for k, v in kw.items()})
    kwg = {k: (g(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    if kernel == "xs":  # the pair called directly (ops.conv1d keeps PRO_NONE and short rows on the fused kernel)
        PRO_KEYS = ("pro", "slope", "stats", "gamma", "beta", "alpha")
        xs = ops.activate(g(x), **{k: v for k, v in kwg.items() if k in PRO_KEYS})
        out = ops.conv1d_xs(xs, wt.to(DEV), C_out, ks, **{k: v for k, v in kwg.items() if k not in PRO_KEYS})
    else:
        out = ops.conv1d(g(x), g(wt) if kernel == "f32" else wt.to(DEV), C_out, ks, **kwg)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    e = rel_err(out, ref)
    assert e < 3e-6, "rel err vs contract %g" % e
    # and against an fp64 evaluation of the un-split operands: fp32 round-off class for all three kernels (the fp32
    # ATen-CPU convolution sits at 0.5e-7 .. 1.2e-6 of the same fp64 reference on these cases)
    e64 = rel_err(out, exact)
    assert e64 < 3e-6, "rel err vs fp64 %g" % e64
    assert ops.status() == 0, "benign inputs must not raise a device-side status bit"


WS_CASES = [
    # (B, C_in, C_out, L, ks, dil): the warp-specialised persistent build of st2_conv1d_f16s (C_out <= 64, AdaIN + Snake)
    (32, 64, 64, 30011, 11, 5),   # 3776 tiles, 14-15 per workgroup, the 64 x 256 layout
    (32, 32, 32, 40001, 7, 3),    # the 32 x 512 layout
    (40, 64, 48, 3001, 3, 1),     # 12 tiles per batch item: every workgroup crosses batch items (parameter-table slots)
    (600, 24, 24, 300, 11, 5),    # one tile (two chunks) per batch item: a new parameter table every tile
    (3, 40, 64, 9001, 7, 1),      # ragged channels; fewer tiles than CUs (forced variant only)
]


@pytest.mark.parametrize("B,C_in,C_out,L,ks,dil", WS_CASES)
@pytest.mark.parametrize("epi", ["plain", "res_stats", "res2_div"])
def test_conv1d_f16s_warp_specialised_build_is_bitwise_the_one_role_kernel(B, C_in, C_out, L, ks, dil, epi, monkeypatch):
    """st2_conv1d_f16s_ws.h stages and multiplies exactly what conv1d_f16s_kernel does (same operands, same MFMA order per
    accumulator, the shared epilogue): outputs and InstanceNorm statistics must be bit-identical, and within the contract's
    tolerance of the oracle."""
    from styletts2_amd import _lib
    lib = _lib.load()
    monkeypatch.setattr(_hooks, "conv_path", "fused")
    gen = torch.Generator().manual_seed(77)
    pitch = (L + 31) // 32 * 32
    x = torch.randn(B, C_in, pitch, generator=gen)[:, :, :L]
    w = torch.randn(C_out, C_in, ks, generator=gen) / math.sqrt(C_in * ks)
    wt = weights.pack_conv_f16s(w)
    bias = torch.randn(C_out, generator=gen)
    h = torch.randn(B, 2 * C_in, generator=gen) * 0.3
    alpha = torch.rand(C_in, generator=gen) + 0.5
    res = torch.randn(B, C_out, pitch, generator=gen)[:, :, :L] if epi != "plain" else None
    res2 = torch.randn(B, C_out, pitch, generator=gen)[:, :, :L] if epi == "res2_div" else None
    xg = torch.empty((B, C_in, pitch), device=DEV).copy_(torch.nn.functional.pad(x, (0, pitch - L)))[:, :, :L]
    st = ops.instnorm_stats(xg)

    def dev_rows(t):
        return None if t is None else torch.empty((B, C_out, pitch), device=DEV).copy_(
            torch.nn.functional.pad(t, (0, pitch - L)))[:, :, :L]

    kw = dict(dil=dil, pad_left=(ks - 1) * dil // 2, bias=g(bias), pro=ops.PRO_ADAIN_SNAKE, stats=st, gamma=g(h)[:, :C_in],
              beta=g(h)[:, C_in:], alpha=g(alpha), res=dev_rows(res), res2=dev_rows(res2), div=3.0 if epi == "res2_div" else 1.0,
              want_stats=epi == "res_stats")
    outs = {}
    try:
        for name, variant in (("one_role", 1), ("ws", 2)):
            lib.st2_conv1d_f16s_set_variant(variant)
            out = torch.empty((B, C_out, pitch), device=DEV)[:, :, :L]
            r = ops.conv1d(xg, wt.to(DEV), C_out, ks, out=out, **kw)
            torch.cuda.synchronize()
            outs[name] = (r[0].clone(), r[1].clone()) if kw["want_stats"] else (r.clone(), None)
    finally:
        lib.st2_conv1d_f16s_set_variant(0)
    assert torch.equal(outs["one_role"][0], outs["ws"][0])
    if kw["want_stats"]:
        assert torch.equal(outs["one_role"][1], outs["ws"][1])
    assert ops.status() == 0
    if B * C_in * L <= 5_000_000:  # the contract itself on the small cases (the large ones ride on the equality above)
        ref = R.conv1d(x, wt, C_out, ks, dil=dil, pad_left=(ks - 1) * dil // 2, bias=bias, pro=R.PRO_ADAIN_SNAKE,
                       stats=st.cpu(), gamma=h[:, :C_in], beta=h[:, C_in:], alpha=alpha, res=res, res2=res2,
                       div=3.0 if epi == "res2_div" else 1.0)
        assert rel_err(outs["ws"][0], ref) < 3e-6


@pytest.mark.parametrize("B,C_in,C_out,L,ks,dil,res", [(2, 128, 128, 2500, 11, 5, True), (1, 256, 256, 515, 3, 1, False),
                                                       (2, 64, 64, 777, 7, 3, True), (1, 32, 22, 1300, 7, 1, False),
                                                       (2, 128, 128, 48001, 11, 1, True),
                                                       # >= 1024 workgroups at 256-column tiles: the 32 x 256 wave-tile
                                                       # builds (k = 7 / 11), two partial-sum tiles per wave
                                                       (6, 128, 128, 48001, 11, 3, True), (16, 256, 256, 8000, 7, 1, True),
                                                       # row-end QUARTER bodies (a last tile whose valid columns fit a quarter
                                                       # of the tile runs the body built with TN / 4 column blocks): their
                                                       # partial sums against the CPU reduction, not only variant vs variant
                                                       (4, 512, 512, 400, 3, 1, True),    # 16 of 128 columns, aligned
                                                       (2, 256, 256, 777, 7, 1, True),    # 9 of 128, unaligned row end
                                                       (16, 256, 256, 8000, 3, 1, True),  # 64 of 256: the wide k = 3 build
                                                       (3, 1024, 1024, 400, 3, 1, False)])
def test_conv1d_xs_epilogue_stats(B, C_in, C_out, L, ks, dil, res, monkeypatch):
    """want_stats: InstanceNorm statistics of the conv OUTPUT from the epilogue's per-tile partial sums
    (st2_conv1d_xs part + st2_stats_finalize) against the fp64 reduction of the stored tensor."""
    monkeypatch.setattr(_hooks, "conv_path", "xs")
    x, w, kw = make_conv_case(seed=99, B=B, C_in=C_in, C_out=C_out, L=L, ks=ks, dil=dil, pro=R.PRO_ADAIN_SNAKE, res=res)
    wt = weights.pack_conv_f16s(w)
    kwg = {k: (g(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    out, st = ops.conv1d(g(x), wt.to(DEV), C_out, ks, want_stats=True, **kwg)
    torch.cuda.synchronize()
    ref = R.conv1d(x, wt, C_out, ks, **kw)
    assert rel_err(out, ref) < 2e-5
    st_ref = R.instnorm_stats(out.cpu())
    assert st.shape == st_ref.shape == (B, C_out, 2)
    assert (st.cpu()[..., 0] - st_ref[..., 0]).abs().max().item() < 2e-6 * max(1.0, st_ref[..., 0].abs().max().item())
    assert ((st.cpu()[..., 1] - st_ref[..., 1]).abs() / st_ref[..., 1]).max().item() < 5e-6


@pytest.mark.parametrize("path,B,C,L", [("xs", 4, 128, 3000), ("xs", 1, 256, 1500), ("fused", 3, 64, 2100), ("fused", 2, 32, 5000),
                                        ("interleave", 2, 64, 3001)])
def test_epilogue_statistics_survive_offset_dominated_channels(path, B, C, L, monkeypatch):
    """InstanceNorm statistics from the producers' per-slot partial sums on channels whose mean is 1 000 x their standard
    deviation (a bias-dominated channel of a residual stream; found on the small-magnitude checkpoints of
    test_calibration_gpu.py, where |mean| / std ~ 100 cost 2e-4 of rstd): the sums are taken of values SHIFTED by the slot's
    first stored value and combined with Chan's formula, so the error stays at fp32 round-off in the units the consuming AdaIN
    sees -- |d mean| * rstd and |d rstd| / rstd -- where E[x^2] - mean^2 from unshifted fp32 sums loses every digit."""
    gen = torch.Generator().manual_seed(5)
    offs = (torch.randn(1, C, 1, generator=gen).sign() * (1.0 + torch.rand(1, C, 1, generator=gen)))  # per-channel offset ~ +-1.5
    if path == "interleave":
        s, p = 5, 0
        Lq = (L + s - 1) // s + 1
        Y = torch.randn(B, s * C, Lq, generator=gen) * 1e-3
        add = offs.expand(B, C, L).contiguous() + torch.randn(B, C, L, generator=gen) * 1e-3
        out, st = ops.convt_interleave(g(Y), C, s, p, L, bias=None, add=g(add), want_stats=True)
    else:
        monkeypatch.setattr(_hooks, "conv_path", "xs" if path == "xs" else "fused")
        ks = 7
        x = torch.randn(B, C, L, generator=gen)
        w = torch.randn(C, C, ks, generator=gen) / math.sqrt(C * ks) * 1e-3
        res = offs.expand(B, C, L).contiguous()
        wt = weights.pack_conv_f16s(w).to(DEV)
        kw = dict(pad_left=3, bias=g(torch.zeros(C)), res=g(res), want_stats=True)
        if path == "xs":
            out, st = ops.conv1d_xs(ops.activate(g(x)), wt, C, ks, **kw)
        else:
            h = torch.randn(B, 2 * C, generator=gen) * 0.3
            out, st = ops.conv1d(g(x), wt, C, ks, pro=ops.PRO_ADAIN_LEAKY, slope=0.2, stats=ops.instnorm_stats(g(x)), gamma=g(h)[:, :C],
                                 beta=g(h)[:, C:], **kw)
    torch.cuda.synchronize()
    y = out.cpu().double()
    mean, var = y.mean(-1), y.var(-1, unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 300.0, "the case is meant to be offset-dominated"
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    # the mean is handed on as an fp32 number (as in the reference): right to one ulp of it; rstd to fp32 round-off
    dm = ((st.cpu()[..., 0].double() - mean).abs() / mean.abs()).max().item()
    dr = ((st.cpu()[..., 1].double() - rstd).abs() / rstd).max().item()
    assert dm < 1.2e-7 and dr < 1e-6, "|d mean| / |mean| = %.2e, |d rstd| / rstd = %.2e" % (dm, dr)


@pytest.mark.parametrize("B,C,L,ks,dil,cols", [(1, 256, 5680, 3, 1, 32),     # 90 tiles of 128 x 128 on 256 CUs -> 32-column tiles
                                                (1, 128, 28400, 3, 1, 64),    # 222 -> 64-column tiles
                                                (2, 128, 8001, 3, 1, 64), (1, 1024, 400, 3, 1, 32),  # 32 tiles, K = 3 072 deep
                                                (1, 128, 40000, 3, 1, 64),    # 313 tiles: still one partial round of the chip
                                                (1, 256, 5680, 7, 1, 32), (1, 128, 9000, 11, 5, 32),  # k = 7 / 11 follow the same rule
                                                (1, 128, 34800, 7, 3, 64),    # ... up to 340 tiles of 128
                                                (1, 128, 48001, 11, 1, 128),  # 376 tiles
                                                (4, 128, 40000, 3, 1, 128)])  # 1 252 tiles: the ordinary build
def test_conv1d_xs_small_grid_builds(B, C, L, ks, dil, cols, monkeypatch):
    """k = 3 / 7 / 11 launches with few 128 x 128 tiles (one utterance: long-form synthesis, BASELINE.json configs[4]) run 64- / 32-column
    tiles by a rule of the geometry (st2_conv1d_xs_part_cols): the output is BITWISE that of the 128-column build
    (same products in the same order per element), the InstanceNorm statistics -- partial sums per 64 / 32 columns instead of
    128 -- meet the CPU reduction of the stored tensor at the same bar, and two runs are bitwise identical."""
    from styletts2_amd import _lib
    monkeypatch.setattr(_hooks, "conv_path", "xs")
    x, w, kw = make_conv_case(seed=41, B=B, C_in=C, C_out=C, L=L, ks=ks, dil=dil, pro=R.PRO_ADAIN_SNAKE, res=True)
    wt_host = weights.pack_conv_f16s(w)
    wt = wt_host.to(DEV)
    kwg = {k: (g(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    PRO_KEYS = ("pro", "slope", "stats", "gamma", "beta", "alpha")
    xs = ops.activate(g(x), **{k: v for k, v in kwg.items() if k in PRO_KEYS})
    ckw = {k: v for k, v in kwg.items() if k not in PRO_KEYS}
    d = _lib.ConvDesc()
    d.B, d.C_in, d.C_out, d.L_in, d.L_out, d.ks = B, C, C, L, L, ks
    assert _lib.load().st2_conv1d_xs_part_cols(d) == cols
    out, st = ops.conv1d_xs(xs, wt, C, ks, want_stats=True, **ckw)
    out2, st2 = ops.conv1d_xs(xs, wt, C, ks, want_stats=True, **ckw)
    wide, st_wide = ops.conv1d_xs(xs, wt, C, ks, want_stats=True, part_cols=128, **ckw)
    plain = ops.conv1d_xs(xs, wt, C, ks, **ckw)  # no statistics: the same rule, the same tiles
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(st, st2), "run-to-run determinism"
    assert torch.equal(out, wide) and torch.equal(out, plain), "tile width must not change a single output bit"
    if B * C * L <= 3_000_000:
        assert rel_err(out, R.conv1d(x, wt_host, C, ks, **kw)) < 2e-5
    if B * C * L > 8_000_000:  # the large case rides on the bitwise equality above; statistics on a slice of the rows
        out, st, st_wide = out[:1], st[:1], st_wide[:1]
    st_ref = R.instnorm_stats(out.cpu())
    for s in (st, st_wide):
        assert (s.cpu()[..., 0] - st_ref[..., 0]).abs().max().item() < 2e-6 * max(1.0, st_ref[..., 0].abs().max().item())
        assert ((s.cpu()[..., 1] - st_ref[..., 1]).abs() / st_ref[..., 1]).max().item() < 5e-6
    assert ops.status() == 0


@pytest.mark.parametrize("pro", [R.PRO_NONE, R.PRO_LEAKY, R.PRO_ADAIN_LEAKY, R.PRO_ADAIN_SNAKE, R.PRO_SNAKE,
                                 R.PRO_COLNORM])
def test_activate_planes_match_contract(pro):
    """st2_act_split: hi + lo planes reconstruct x_scale * pro(x) (x_scale = ops.x_scale_for(pro)); halo, tail and channel padding are exact zeros; every
    stored half is finite and |lo| <= half an ulp of hi."""
    B, C, L = 2, 70, 333
    x, w, kw = make_conv_case(seed=5, B=B, C_in=C, C_out=8, L=L, ks=1, dil=1, pro=pro)
    akw = {k: (g(v) if torch.is_tensor(v) else v) for k, v in kw.items()
           if k in ("pro", "slope", "stats", "gamma", "beta", "alpha")}
    xs = ops.activate(g(x), **akw)
    torch.cuda.synchronize()
    d = xs.data.cpu().float()                                   # [B, 2, cg, Lp, 8]
    assert xs.C == C and xs.L == L and d.shape[2] * 8 >= C and d.shape[3] >= L + xs.halo
    val = (d[:, 0] + d[:, 1]).permute(0, 1, 3, 2).reshape(B, -1, d.shape[3]) / xs.x_scale   # [B, cg*8, Lp]
    assert xs.x_scale == ops.x_scale_for(pro) == (8.0 if pro in (R.PRO_ADAIN_LEAKY, R.PRO_ADAIN_SNAKE, R.PRO_COLNORM) else 1.0)
    ref = R.activate(x, **{k: v for k, v in kw.items() if k in ("pro", "slope", "stats", "gamma", "beta", "alpha")})
    got = val[:, :C, xs.halo:xs.halo + L]
    assert (got - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
    assert val[:, :, :xs.halo].abs().max().item() == 0.0 and val[:, :, xs.halo + L:].abs().max().item() == 0.0
    assert val[:, C:].abs().max().item() == 0.0
    assert bool(torch.isfinite(d).all())


def test_activate_segment_affine_matches_contract():
    """st2_act_split with gb_seg: the token-merged view [1, C, G * N] of G utterances whose LayerNorm affine is per
    utterance -- row l // N of gamma / beta applies at column l -- and the k = 1 conv over it equals the per-utterance
    fused conv on the [G, C, N] view of the same storage."""
    G, C, N, C_out = 5, 160, 61, 40  # C > 128: a k = 1 conv with a prologue takes the act_split + xs pair (ops.prefer_fused)
    gen_ = torch.Generator().manual_seed(11)
    store = torch.randn(C, G, N, generator=gen_)                       # token-merged channel-major storage
    xb = store.permute(1, 0, 2)                                         # [G, C, N] view
    xm = store.reshape(1, C, G * N)                                     # [1, C, G*N] view
    gamma, beta = torch.randn(G, C, generator=gen_) * 0.3, torch.randn(G, C, generator=gen_) * 0.3
    st = R.colnorm_stats(xb)                                            # [G, N, 2]
    stm = st.reshape(1, G * N, 2)
    ref = R.activate(xm, pro=R.PRO_COLNORM, stats=stm, gamma=gamma, beta=beta, gamma_plus_one=True, gb_seg=N)
    per = torch.cat([R.activate(xb[i:i + 1], pro=R.PRO_COLNORM, stats=st[i:i + 1], gamma=gamma[i:i + 1],
                                beta=beta[i:i + 1], gamma_plus_one=True) for i in range(G)], dim=2)
    assert torch.equal(ref, per)                                        # the contract itself: segment rows == per-utterance
    xs = ops.activate(g(store).reshape(1, C, G * N), pro=R.PRO_COLNORM, stats=g(stm), gamma=g(gamma), beta=g(beta),
                      gamma_plus_one=True, gb_seg=N)
    torch.cuda.synchronize()
    d = xs.data.cpu().float()
    val = (d[:, 0] + d[:, 1]).permute(0, 1, 3, 2).reshape(1, -1, d.shape[3]) / xs.x_scale
    got = val[:, :C, xs.halo:xs.halo + G * N]
    assert (got - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
    w = torch.randn(C_out, C, 1, generator=gen_) * 0.1
    wt = weights.pack_conv_f16s(w).to(DEV)
    y_m = ops.conv1d(g(store).reshape(1, C, G * N), wt, C_out, 1, pro=R.PRO_COLNORM, stats=g(stm), gamma=g(gamma),
                     beta=g(beta), gamma_plus_one=True, gb_seg=N)       # G * N >= 256: act_split + xs conv
    y_b = ops.conv1d(g(store).permute(1, 0, 2), wt, C_out, 1, pro=R.PRO_COLNORM, stats=g(st), gamma=g(gamma),
                     beta=g(beta), gamma_plus_one=True)                 # N < 256: fused kernel per utterance
    ref_y = R.conv1d(xm, weights.pack_conv_f16s(w), C_out, 1, pro=R.PRO_COLNORM, stats=stm, gamma=gamma, beta=beta,
                     gamma_plus_one=True, gb_seg=N)
    assert rel_err(y_m, ref_y) < 3e-6
    assert rel_err(y_b.permute(1, 0, 2).reshape(1, C_out, G * N), ref_y) < 3e-6


def test_conv1d_writes_into_channel_slice():
    """Producers write straight into the [x | asr_res | F0 | N] concat buffer (Modules/istftnet.py:522)."""
    x, w, kw = make_conv_case(seed=7, B=2, C_in=16, C_out=24, L=70, ks=3, dil=1, pro=R.PRO_NONE)
    wt = weights.pack_conv(w)
    ref = R.conv1d(x, wt, 24, 3, **kw)
    big_in = torch.full((2, 40, 70), 7.0, device=DEV)
    big_in[:, 5:21] = g(x)
    big_out = torch.full((2, 50, 70), -3.0, device=DEV)
    kwg = {k: (g(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    ops.conv1d(big_in[:, 5:21], g(wt), 24, 3, out=big_out[:, 10:34], **kwg)
    torch.cuda.synchronize()
    assert rel_err(big_out[:, 10:34], ref) < 2e-5
    assert (big_out[:, :10] == -3.0).all() and (big_out[:, 34:] == -3.0).all()


def test_conv1d_rejects_bad_arguments():
    from styletts2_amd._lib import St2Error
    x = torch.randn(1, 4, 8, device=DEV)
    wt = weights.pack_conv(torch.randn(4, 4, 4)).to(DEV)
    with pytest.raises(St2Error):
        ops.conv1d(x, wt, 4, 4)  # unsupported kernel size
    with pytest.raises(St2Error):
        ops.conv1d(x.cpu(), wt, 4, 4)  # no CPU path


@pytest.mark.parametrize("B,C,L", [(2, 5, 7), (3, 66, 400), (2, 128, 48001), (1, 3, 2049)])
def test_instnorm_stats(B, C, L):
    gen = torch.Generator().manual_seed(L)
    x = torch.randn(B, C, L, generator=gen) * 2.0 + 5.0
    ref = R.instnorm_stats(x)
    out = ops.instnorm_stats(g(x))
    assert rel_err(out[..., 0], ref[..., 0]) < 1e-6
    assert rel_err(out[..., 1], ref[..., 1]) < 1e-5
    out2 = ops.instnorm_stats(g(x))
    assert torch.equal(out, out2), "reduction must be bitwise reproducible"


def test_colnorm_stats():
    x = torch.randn(3, 1024, 100) + 0.5
    ref = R.colnorm_stats(x)
    out = ops.colnorm_stats(g(x))
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("B,K,J,act", [(1, 128, 300, R.ACT_NONE), (32, 128, 4100, R.ACT_NONE),
                                       (9, 257, 1024, R.ACT_GELU), (3, 1024, 1024, R.ACT_GELU)])
def test_style_fc(B, K, J, act):
    gen = torch.Generator().manual_seed(J)
    s = torch.randn(B, K, generator=gen)
    wt = torch.randn(K, J, generator=gen) / math.sqrt(K)
    bias = torch.randn(J, generator=gen)
    ref = R.style_fc(s, wt, bias, act)
    out = ops.style_fc(g(s), g(wt), g(bias), act)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("C_in,C_out,s,p,op,L,reflect", [(16, 8, 10, 5, 0, 33, False), (12, 6, 6, 3, 0, 50, True),
                                                        (8, 8, 5, 3, 1, 21, False), (8, 4, 3, 2, 1, 19, False),
                                                        (6, 4, 2, 1, 0, 40, False), (8, 6, 6, 3, 0, 517, True),
                                                        (4, 5, 10, 5, 0, 311, False), (4, 3, 2, 1, 0, 1700, False)])
def test_conv_transpose_polyphase(C_in, C_out, s, p, op, L, reflect):
    """ups[i] of both vocoders: polyphase GEMM + interleave == ConvTranspose1d (+ ReflectionPad1d((1,0)))."""
    gen = torch.Generator().manual_seed(s)
    x = torch.randn(2, C_in, L, generator=gen)
    w = torch.randn(C_in, C_out, 2 * s, generator=gen) * 0.1
    b = torch.randn(C_out, generator=gen)
    ref = torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x, 0.1), w, b, stride=s, padding=p,
                                               output_padding=op)
    L_raw = ref.shape[2]
    if reflect:
        ref = torch.nn.functional.pad(ref, (1, 0), mode="reflect")
    add = torch.randn(ref.shape, generator=gen)
    ref = ref + add
    wt = weights.pack_conv(weights.polyphase_convt(w, s))
    Y = ops.conv1d(g(x), g(wt), s * C_out, 2, pad_left=1, L_out=L + 1, pro=R.PRO_LEAKY, slope=0.1)
    out = ops.convt_interleave(Y, C_out, s, p, L_raw, bias=g(b), add=g(add), reflect_left=reflect)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-5
    # the same with the InstanceNorm statistics of the output from the kernel's per-tile partial sums
    out2, st = ops.convt_interleave(Y, C_out, s, p, L_raw, bias=g(b), add=g(add), reflect_left=reflect, want_stats=True)
    assert torch.equal(out2, out)
    st_ref = R.instnorm_stats(out.cpu())
    assert (st.cpu()[..., 0] - st_ref[..., 0]).abs().max().item() < 2e-6 * max(1.0, st_ref[..., 0].abs().max().item())
    assert ((st.cpu()[..., 1] - st_ref[..., 1]).abs() / st_ref[..., 1]).max().item() < 5e-6


def test_conv1d_direct():
    gen = torch.Generator().manual_seed(3)
    for (C_in, C_out, ks, st, pad, L) in [(22, 16, 12, 6, 3, 481), (22, 8, 1, 1, 0, 100), (1, 1, 3, 2, 1, 80),
                                          (1, 8, 60, 30, 15, 3000), (256, 1, 1, 1, 0, 55)]:
        x = torch.randn(2, C_in, L, generator=gen)
        w = torch.randn(C_out, C_in, ks, generator=gen) * 0.2
        b = torch.randn(C_out, generator=gen)
        ref = R.conv1d_direct(x, w, b, st, pad)
        out = ops.conv1d_direct(g(x), g(w), g(b), st, pad)
        assert out.shape == ref.shape
        assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("kernel", ["f32", "f16s"])
@pytest.mark.parametrize("C_in,C_out,stride,L", [(22, 256, 6, 4801), (1, 40, 30, 6000), (1, 64, 2, 301), (3, 8, 5, 77)])
def test_strided_conv_as_phase_split_plus_k2_conv(C_in, C_out, stride, L, kernel):
    """noise_convs of both vocoders (kernel = 2*stride, padding = (stride+1)//2): st2_phase_split + a stride-1 k=2
    conv over C_in*stride channels == F.conv1d(stride=stride) (Modules/istftnet.py:332-336,361)."""
    gen = torch.Generator().manual_seed(stride)
    x = torch.randn(2, C_in, L, generator=gen)
    w = torch.randn(C_out, C_in, 2 * stride, generator=gen) * 0.2
    b = torch.randn(C_out, generator=gen)
    pad = (stride + 1) // 2
    ref = torch.nn.functional.conv1d(x, w, b, stride=stride, padding=pad)
    L_out = ref.shape[2]
    xp = ops.phase_split(g(x), stride, pad, L_out + 1)
    assert torch.equal(xp.cpu(), R.phase_split(x, stride, pad, L_out + 1))
    w2 = weights.polyphase_strided_conv(w, stride)
    wt = g(weights.pack_conv(w2)) if kernel == "f32" else weights.pack_conv_f16s(w2).to(DEV)
    out = ops.conv1d(xp, wt, C_out, 2, pad_left=0, L_out=L_out, bias=g(b))
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-5


def test_adain_leaky_pool():
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 70, 33, generator=gen) + 1.0
    st = R.instnorm_stats(x)
    h = torch.randn(2, 140, generator=gen) * 0.3
    w = torch.randn(70, 3, generator=gen)
    b = torch.randn(70, generator=gen)
    ref = R.adain_leaky_pool(x, st, h[:, :70], h[:, 70:], 0.2, w, b)
    hg = g(h)
    out = ops.adain_leaky_pool(g(x), g(st), hg[:, :70], hg[:, 70:], 0.2, g(w), g(b))
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("F,U", [(16, 300), (800, 300)])
def test_har_source_bit_faithful_phase(F, U):
    """SineGen phases reach ~1e5 rad; one fp32 ulp there moves sin() by ~8e-3, so this passes only if the
    cumsum/interp op order matches ATen-CPU.  Includes unvoiced and negative F0."""
    gen = torch.Generator().manual_seed(F)
    B, H = 2, 9
    f0 = torch.rand(B, F, generator=gen) * 300.0 + 80.0
    f0[0, : F // 4] = 0.0
    f0[1, F // 2: F // 2 + 3] = -40.0
    noise = torch.randn(B, F * U, H, generator=gen)
    lw = torch.randn(H, generator=gen) * 0.5
    lb = torch.randn(1, generator=gen) * 0.1
    ref = R.har_source(f0, U, noise, lw, lb)
    out = ops.har_source(g(f0), U, g(noise), g(lw), g(lb)).cpu()
    diff = (out - ref).abs()
    assert diff.max().item() < 2e-5, "max %g, frac>1e-4: %g" % (diff.max().item(), (diff > 1e-4).float().mean().item())


def test_stft_mag_and_phase_mod_2pi():
    gen = torch.Generator().manual_seed(11)
    x = torch.tanh(torch.randn(2, 4000, generator=gen))
    ref = R.stft_mag_phase(x, 20, 5)
    out = ops.stft_mag_phase(g(x), 20, 5).cpu()
    assert out.shape == ref.shape
    assert (out[:, :11] - ref[:, :11]).abs().max().item() < 2e-5
    # phase is ill-conditioned where |X| ~ 0 and wraps at +-pi: compare on the unit circle, weighted by magnitude
    d = torch.remainder(out[:, 11:] - ref[:, 11:] + math.pi, 2 * math.pi) - math.pi
    assert (d.abs() * ref[:, :11]).max().item() < 5e-5


def test_istft():
    gen = torch.Generator().manual_seed(13)
    M = 801
    sp = torch.cat([torch.exp(torch.randn(2, 11, M, generator=gen)), torch.sin(torch.randn(2, 11, M, generator=gen) * 3)], 1)
    ref = R.istft(sp, 20, 5)
    out = ops.istft(g(sp), 20, 5).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("B,N", [(2, 100), (1, 5), (2, 190), (1, 512)])
def test_attention(B, N):
    gen = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, 3 * 512, N, generator=gen)
    q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
    ref = R.attention(q, k, v, 8, 64 ** -0.5)
    t = g(qkv)
    out = ops.attention(t[:, :512], t[:, 512:1024], t[:, 1024:], 8, 64 ** -0.5)
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("B,N", [(3, 100), (2, 37)])
def test_attention_key_padding(B, N):
    """st2_attention_keylen: keys past key_len[b] are excluded from every query's softmax (HF additive -inf mask)."""
    gen = torch.Generator().manual_seed(N)
    q, k, v = (torch.randn(B, 12 * 64, N, generator=gen) for _ in range(3))
    lens = torch.tensor([N, max(1, N // 3), 1][:B], dtype=torch.int32)
    ref = R.attention(q, k, v, 12, 0.125, key_len=lens)
    out = ops.attention(g(q), g(k), g(v), 12, 0.125, key_len=lens.to(DEV))
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("plus_one,act,ragged", [(False, R.ACT_NONE, False), (True, R.ACT_NONE, True),
                                                 (False, R.ACT_LEAKY, True)])
def test_colnorm_apply(plus_one, act, ragged):
    gen = torch.Generator().manual_seed(3)
    B, C, L = 3, 70, 301
    x = torch.randn(B, C, L, generator=gen) * 2 + 0.5
    st = R.colnorm_stats(x, eps=1e-12)
    gamma = torch.randn(B if plus_one else 1, C, generator=gen)
    beta = torch.randn(B if plus_one else 1, C, generator=gen)
    lens = torch.tensor([L, 17, 200], dtype=torch.int32) if ragged else None
    ref = R.colnorm_apply(x, st, gamma, beta, gamma_plus_one=plus_one, act=act, slope=0.2, lengths=lens)
    big = torch.full((B, C + 6, L), -7.0, device=DEV)
    out = ops.colnorm_apply(g(x), ops.colnorm_stats(g(x), eps=1e-12), g(gamma), g(beta), gamma_plus_one=plus_one,
                            act=act, slope=0.2, lengths=None if lens is None else lens.to(DEV), out=big[:, 3:3 + C])
    assert rel_err(out, ref) < 1e-5
    assert (big[:, :3] == -7.0).all() and (big[:, 3 + C:] == -7.0).all()
    if ragged:
        assert float(out[1, :, 17:].abs().max()) == 0.0


def test_plbert_engine_matches_hf():
    """PL-BERT on the engine's kernels (GPU) against the HF AlbertModel forward on CPU, padded batch."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _util import manifest
    from transformers import AlbertModel
    from styletts2_amd import models
    from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
    bert = models.load_plbert(manifest("ljspeech")["plbert"]).eval()
    synth.init_synthetic_(bert, 15)
    B, N = 4, 100
    gen = torch.Generator().manual_seed(0)
    ids = torch.randint(1, 178, (B, N), generator=gen)
    ids[:, 0] = 0
    lens = torch.tensor([100, 61, 100, 7])
    mask = (torch.arange(N).unsqueeze(0) < lens.unsqueeze(1)).int()
    with torch.no_grad():
        ref = AlbertModel.forward(bert, ids, attention_mask=mask).last_hidden_state
    bert = bert.to(DEV)
    out = bert(g(ids), attention_mask=g(mask))
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert (out.cpu() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


def test_token_glue():
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(3, 40, 101, generator=gen)
    v = torch.randn(3, 40, generator=gen)
    assert rel_err(ops.add_chanvec(g(x), g(v)), R.add_chanvec(x, v)) < 1e-6
    assert rel_err(ops.mean_tokens(g(x)), R.mean_tokens(x)) < 1e-6
    y, z = torch.randn(3, 40, 101, generator=gen), torch.randn(3, 40, 101, generator=gen)
    assert rel_err(ops.axpbypcz(g(x), 0.3, g(y), -1.2, g(z), 2.0), R.axpbypcz(x, 0.3, y, -1.2, z, 2.0)) < 1e-6
    assert rel_err(ops.axpbypcz(g(x), 0.3, g(y), -1.2), R.axpbypcz(x, 0.3, y, -1.2)) < 1e-6


def _conv_vs_fp64(x, w, pro, path, monkeypatch, **extra):
    """One conv (C_in x ks x C_out from w) on the engine vs an fp64 evaluation of the un-split operands."""
    monkeypatch.setattr(_hooks, "conv_path", "xs" if path == "xs" else "fused")
    C_out, C_in, ks = w.shape
    x_scale = extra.pop("x_scale", None)  # a calibrated operand scale: engine only, the fp64 evaluation has none
    kw = dict(pad_left=(ks - 1) // 2, pro=pro, **extra)
    if pro == R.PRO_LEAKY:
        kw["slope"] = 0.1
    exact = R.conv1d(x.double(), weights.pack_conv(w).double(), C_out, ks, **kw)
    wt = weights.pack_conv_f16s(w).to(DEV)
    kw["x_scale"] = x_scale
    if path == "xs":
        PRO_KEYS = ("pro", "slope", "x_scale")
        xs = ops.activate(g(x), **{k: v for k, v in kw.items() if k in PRO_KEYS})
        out = ops.conv1d_xs(xs, wt, C_out, ks, **{k: v for k, v in kw.items() if k not in PRO_KEYS})
    else:
        out = ops.conv1d(g(x), wt, C_out, ks, **kw)
    torch.cuda.synchronize()
    return out.cpu().double(), exact


@pytest.mark.parametrize("path", ["xs", "fused"])
@pytest.mark.parametrize("mag", [1e-4, 1e-3, 1e-2, 1.0, 1e3, 1e4])
@pytest.mark.parametrize("pro", [R.PRO_NONE, R.PRO_LEAKY])
def test_split_f16_conv_dynamic_range(mag, pro, path, monkeypatch):
    """The reference's convs are fp32 at every magnitude (Modules/istftnet.py:68-74).  With the layer's operand scale
    calibrated to its input -- x_scale = 2^floor(log2(8192 / max |pro(x)|)), what st2_calibrate installs per conv site --
    both f16 halves of every significant operand are normal numbers and the conv is held to ONE bar, 3e-6 of the output
    maximum against fp64, from |x| ~ 1e-4 to 1e4."""
    ops.status(clear=True)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 96, 700, generator=gen) * mag
    x[0, 3, 100] = 4.0 * mag  # an outlier well above the bulk
    w = torch.randn(80, 96, 3, generator=gen) / math.sqrt(96 * 3)
    top = R.activate(x, pro=pro, slope=0.1).abs().max().item() if pro == R.PRO_LEAKY else x.abs().max().item()
    xsc = ops.calibrated_x_scale(top)
    assert 4096.0 <= top * xsc < 8192.0
    out, exact = _conv_vs_fp64(x, w, pro, path, monkeypatch, x_scale=xsc)
    assert bool(torch.isfinite(out).all())
    e = ((out - exact).abs().max() / exact.abs().max()).item()
    assert e < 3e-6, "rel err vs fp64 %g at |x| ~ %g (x_scale %g)" % (e, mag, xsc)
    assert ops.status() == 0


@pytest.mark.parametrize("path", ["xs", "fused"])
@pytest.mark.parametrize("mag", [1e-3, 1.0, 1e3, 1e4])
def test_split_f16_conv_dynamic_range_by_rule(mag, path, monkeypatch):
    """The same conv WITHOUT calibration (x_scale = 1 for an un-normalised input): |x| up to 65504 stays exact to fp32
    round-off, but the lo half bottoms out in the f16 subnormals, so the operand's precision is max(2^-22 relative, 2^-25
    ABSOLUTE): O(1) and above meet 3e-6, a tensor at 1e-3 only ~1e-5 ... 1e-4.  This is the floor st2_debug_headroom's
    rel_err column reports and the reason a serving process calibrates (pipeline.calibrate)."""
    ops.status(clear=True)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 96, 700, generator=gen) * mag
    w = torch.randn(80, 96, 3, generator=gen) / math.sqrt(96 * 3)
    out, exact = _conv_vs_fp64(x, w, R.PRO_NONE, path, monkeypatch)
    e = ((out - exact).abs().max() / exact.abs().max()).item()
    assert e < (3e-6 if mag >= 1.0 else 1e-4), "rel err vs fp64 %g at |x| ~ %g" % (e, mag)
    assert ops.status() == 0


@pytest.mark.parametrize("path", ["xs", "fused"])
def test_split_f16_conv_saturates_and_reports_instead_of_nan(path, monkeypatch):
    """Beyond the f16 range the operand is clamped to +-65504 (never inf / NaN) and ST2_STATUS_F16_RANGE is raised;
    ops.check_status() turns it into an exception and clears it."""
    from styletts2_amd import _lib
    ops.status(clear=True)
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(1, 64, 600, generator=gen)
    x[0, 5, 300] = 3.0e5
    x[0, 9, 17] = -1.0e6
    w = torch.randn(64, 64, 3, generator=gen) / math.sqrt(64 * 3)
    out, exact = _conv_vs_fp64(x, w, R.PRO_NONE, path, monkeypatch)
    assert bool(torch.isfinite(out).all()), "a clamped operand must not produce inf / NaN"
    assert ops.status() & _lib.STATUS_F16_RANGE
    with pytest.raises(_lib.St2Error, match="f16 range"):
        ops.check_status()
    assert ops.status() == 0, "check_status() clears the word"
    # positions the outliers do not reach are unaffected
    far = torch.ones(600, dtype=torch.bool)
    far[298:303] = False
    far[15:20] = False
    assert ((out - exact)[:, :, far].abs().max() / exact[:, :, far].abs().max()).item() < 3e-6


@pytest.mark.parametrize("path", ["xs", "fused"])
def test_split_f16_conv_per_row_weight_spread(path, monkeypatch):
    """Output rows whose weights differ by 2^20 (weight-norm gains of a trained checkpoint are free parameters): the
    packing scales every output row by its own power of two, so EVERY row -- not just the loudest -- meets the fp32
    round-off bar relative to its own magnitude."""
    ops.status(clear=True)
    gen = torch.Generator().manual_seed(9)
    C_out = 128
    x = torch.randn(2, 128, 900, generator=gen) * 1.5 + 0.3
    w = torch.randn(C_out, 128, 7, generator=gen) / math.sqrt(128 * 7)
    w = w * torch.pow(2.0, -torch.arange(C_out).float() * 20.0 / (C_out - 1)).view(-1, 1, 1)
    out, exact = _conv_vs_fp64(x, w, R.PRO_NONE, path, monkeypatch)
    per_row = ((out - exact).abs().amax(dim=(0, 2)) / exact.abs().amax(dim=(0, 2)))
    assert per_row.max().item() < 3e-6, "worst row rel err %g (row %d)" % (per_row.max().item(), int(per_row.argmax()))
    assert ops.status() == 0


def test_glue_kernels():
    gen = torch.Generator().manual_seed(19)
    w = torch.randn(128, generator=gen)
    assert rel_err(ops.time_features(-1.37, g(w), 5), R.time_features(-1.37, w, 5)) < 1e-6
    e = torch.randn(3, 37, 70, generator=gen)
    buf = torch.zeros(3, 90, 37, device=DEV)
    ops.tokens_to_channels(g(e), buf[:, 20:])
    assert torch.equal(buf[:, 20:].cpu(), e.transpose(1, 2)) and float(buf[:, :20].abs().max()) == 0.0
    merged = torch.zeros(70, 3, 37, device=DEV).permute(1, 0, 2)  # the denoiser's token-merged layout
    ops.tokens_to_channels(g(e[0].contiguous()), merged, B=3)     # [N, E] table broadcast over the batch
    assert all(torch.equal(merged[b].cpu(), e[0].t()) for b in range(3))
    x = torch.randn(4, 33, generator=gen)
    out = torch.zeros(4, 40, 21, device=DEV)
    ops.broadcast_cols(g(x), out[:, :33])
    assert torch.equal(out[:, :33].cpu(), x.unsqueeze(-1).expand(4, 33, 21)) and float(out[:, 33:].abs().max()) == 0.0
    y = torch.randn(2, 9, 300, generator=gen)
    dst = torch.zeros(2, 12, 300, device=DEV)
    ops.copy_ncl(g(y)[:, 2:7], dst[:, 5:10])
    assert torch.equal(dst[:, 5:10].cpu(), y[:, 2:7])


@pytest.mark.parametrize("shift", [False, True])
def test_expand_by_durations_is_the_one_hot_matmul(shift):
    gen = torch.Generator().manual_seed(20)
    B, C, N = 3, 130, 57
    x = torch.randn(B, C, N, generator=gen)
    dur = torch.randint(0, 9, (B, N), generator=gen)
    dur[:, 0] = 1
    T = int(dur.sum(dim=1).max())
    for b in range(B):  # equal row sums: top every row up on its last token
        dur[b, -1] += T - int(dur[b].sum())
    out = ops.expand_by_durations(g(x), g(dur), T, shift=shift)
    assert torch.equal(out.cpu(), R.expand_by_durations(x, dur, T, shift=shift))  # a gather: bit-exact
    aln = torch.zeros(B, N, T)
    for b in range(B):
        c = 0
        for i in range(N):
            aln[b, i, c:c + int(dur[b, i])] = 1
            c += int(dur[b, i])
    assert torch.equal(R.expand_by_durations(x, dur, T), x @ aln)
    torch.cuda.synchronize()
    assert ops.status() == 0
    if not shift:  # a frame count that is not the durations' row sum is reported, never silent
        ops.expand_by_durations(g(x), g(dur), T + 3, shift=False)
        torch.cuda.synchronize()
        with pytest.raises(Exception, match="durations does not sum"):
            ops.check_status()
        assert ops.status() == 0  # cleared by the check


def test_duration_head_matches_contract():
    """Linear 512 -> 50 + sigmoid sum + round + clamp + pad masking + tail in one kernel vs the torch ops of the
    notebook (ipynb:296-301): the un-rounded sums to fp32 round-off, the integer durations exactly wherever the sum is
    not within 1e-4 of a rounding tie."""
    gen = torch.Generator().manual_seed(21)
    B, K, J, N = 4, 512, 50, 83
    x = torch.randn(B, K, N, generator=gen) * 0.5
    w = torch.randn(J, K, generator=gen) / math.sqrt(K)
    bias = torch.randn(J, generator=gen) * 0.1
    lens = torch.tensor([83, 40, 1, 82], dtype=torch.int32)
    for lengths, tail in ((None, 0), (lens, 5)):
        dur, sums = ops.duration_head(g(x), g(w), g(bias), lengths=g(lengths), tail=tail, want_sums=True)
        rd, rs = R.duration_head(x, w, bias, lengths=lengths, tail=tail, want_sums=True)
        assert (sums.cpu() - rs).abs().max().item() < 2e-5
        safe = ((rs - rs.floor() - 0.5).abs() > 1e-4)
        assert torch.equal(dur.cpu()[safe], rd[safe]) and dur.dtype == torch.int64
        if lengths is not None:
            assert int(dur.cpu()[1, 40:].sum()) == 0 and int(dur[2, 0]) >= 6


def test_mean_tokens_with_lengths():
    gen = torch.Generator().manual_seed(18)
    x = torch.randn(4, 40, 57, generator=gen)
    lens = torch.tensor([57, 1, 30, 56], dtype=torch.int32)
    ref = R.mean_tokens(x, lengths=lens)
    assert rel_err(ops.mean_tokens(g(x), lengths=g(lens)), ref) < 1e-6


def test_lstm_coop_timeout_is_reported_not_silent(monkeypatch):
    """A cooperative group whose partners do not show up in time must not return garbage silently: with the poll
    budget forced to 1 the hand-off fails, ST2_STATUS_LSTM_TIMEOUT is raised and ops.check_status() throws."""
    from styletts2_amd import _lib
    from styletts2_amd.text import EngineLSTM
    monkeypatch.setattr(_hooks, "lstm", "coop")
    monkeypatch.setattr(_hooks, "lstm_recover", False)  # the bare cooperative launch (kernel-level callers)
    ops.status(clear=True)
    lib = _lib.load()
    torch.manual_seed(3)
    lstm = EngineLSTM(640, 256).to(DEV)
    x = torch.randn(8, 640, 50, device=DEV)
    lib.st2_lstm_coop_set_spin_limit(1)
    try:
        lstm.forward_cm(x)
        torch.cuda.synchronize()
    finally:
        lib.st2_lstm_coop_set_spin_limit(0)
    assert ops.status() & _lib.STATUS_LSTM_TIMEOUT
    assert ops.lstm_coop_status() == 1
    with pytest.raises(_lib.St2Error, match="BiLSTM"):
        ops.check_status()
    y = lstm.forward_cm(x)  # and the next call, with the default budget, is fine again
    torch.cuda.synchronize()
    assert ops.status() == 0 and bool(torch.isfinite(y).all())


@pytest.mark.parametrize("plan", ["python", "engine"])
def test_lstm_coop_timeout_is_recovered_in_stream(plan, monkeypatch):
    """What the launch plans issue (st2_lstm_bidir_coop_recovering): with the poll budget forced to 1 every cooperative group
    times out, the conditional single-CU kernel queued behind it re-runs the call, and the caller gets that kernel's outputs
    -- bitwise -- with ST2_STATUS_LSTM_RECOVERED (a warning), not ST2_STATUS_LSTM_TIMEOUT (an exception) and not garbage.
    `engine`: the same through a C++ plan (st2_text_forward: embedding -> convs -> BiLSTM)."""
    import warnings
    from styletts2_amd import _lib
    lib = _lib.load()
    ops.status(clear=True)
    torch.manual_seed(3)
    if plan == "python":
        from styletts2_amd.text import EngineLSTM
        lstm = EngineLSTM(640, 256).to(DEV)
        x = torch.randn(8, 640, 50, device=DEV)
        run = lambda: lstm.forward_cm(x)
    else:
        from _util import manifest
        from styletts2_amd import engine, models
        from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
        man = manifest("ljspeech")
        te = models.build_model(models.recursive_munch(man["config"]), None, None, models.load_plbert(man["plbert"])).text_encoder
        synth.init_synthetic_(te, 4)
        eng = engine.build_text_engine(te.eval(), torch.device(DEV, torch.cuda.current_device()))
        tokens = torch.randint(1, 178, (6, 40), device=DEV)
        run = lambda: eng.text_forward(tokens)
    with _hooks.override(lstm="single"):
        lib.st2_lstm_coop_set_block(-1)  # the C++ plans follow the library hook: no cooperative launches
        try:
            want = run()
            torch.cuda.synchronize()
        finally:
            lib.st2_lstm_coop_set_block(0)
    assert ops.status() == 0
    lib.st2_lstm_coop_set_spin_limit(1)
    try:
        got = run()
        torch.cuda.synchronize()
    finally:
        lib.st2_lstm_coop_set_spin_limit(0)
    st = ops.status()
    assert st & _lib.STATUS_LSTM_RECOVERED and not st & _lib.STATUS_LSTM_TIMEOUT, hex(st)
    assert torch.equal(got, want), "the recovered call returns the single-CU kernel's outputs"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ops.check_status()  # informational: no exception
    assert any("single-CU" in str(x.message) for x in w) and ops.status() == 0
    again = run()  # default budget: the cooperative kernel's own outputs, nothing to recover
    torch.cuda.synchronize()
    assert ops.status() == 0
    assert (again - want).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,N", [(1, 96), (32, 100)])
def test_lstm_coop_scratch_is_cleared_on_every_graph_replay(B, N):
    """The cooperative launch clears its scratch (status word, counters, granule tags) at the start of EVERY call, also when the
    call is a node sequence of a replayed hipGraph.  It used to do so with hipMemsetAsync, whose recorded node zeroes on the first
    replay only (ROCm 7.2: later replays leave an 8-byte pointer-like pattern at the head of the buffer) -- invisible until the
    in-stream recovery started reading scratch[0]: every replayed front then re-ran its BiLSTMs on the single-CU kernel (+27 ms
    per long-form passage, profiles/r05/r05d_*).  Now a kernel node: scratch[0] == 0, no status bit and bitwise the eager outputs on
    replays 1..3 of a buffer pre-filled with garbage."""
    from styletts2_amd import _lib
    lib = _lib.load()
    H = 256
    gen = torch.Generator().manual_seed(11)
    whh = g(torch.randn(2, H, 4 * H, generator=gen) / 16).contiguous()
    G = g(torch.randn(B, 8 * H, N, generator=gen))
    nbytes = lib.st2_lstm_coop_scratch_bytes(B)
    assert nbytes > 0
    Y = torch.empty(B, 2 * H, N, device=DEV)
    scratch = torch.full((nbytes,), 0x5A, device=DEV, dtype=torch.uint8)

    def call():
        rc = lib.st2_lstm_bidir_coop_recovering(G.data_ptr(), G.stride(0), G.stride(1), whh.data_ptr(), 0, B, H, N, Y.data_ptr(),
                                                Y.stride(0), Y.stride(1), scratch.data_ptr(), nbytes,
                                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.st2_last_error()
    call()
    torch.cuda.synchronize()
    ref = Y.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for it in range(3):
        scratch.fill_(0x5A)
        Y.fill_(7.0)
        ops.status(clear=True)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert int(scratch[:4].view(torch.int32).item()) == 0, "replay %d: scratch[0] not cleared" % it
        assert ops.status(clear=True) == 0, "replay %d raised a status bit" % it
        assert torch.equal(Y, ref), "replay %d differs from the eager call" % it


@pytest.mark.parametrize("xch", [0, 1, 2])
def test_lstm_coop_exchange_variants_agree(xch):
    """The three hand-off forms of st2_lstm_bidir_coop (fences + counter, sc1 + counter, tagged 8-byte granules) are
    the same arithmetic: bitwise equal outputs."""
    from styletts2_amd import _lib
    from styletts2_amd.text import EngineLSTM
    lib = _lib.load()
    torch.manual_seed(5)
    lstm = EngineLSTM(640, 256).to(DEV)
    x = torch.randn(9, 640, 77, device=DEV)
    lens = torch.tensor([77, 5, 40, 77, 76, 1, 33, 60, 77], dtype=torch.int32, device=DEV)
    lib.st2_lstm_coop_set_exchange(2)
    ref = lstm.forward_cm(x, lens)
    lib.st2_lstm_coop_set_exchange(xch)
    try:
        out = lstm.forward_cm(x, lens)
        torch.cuda.synchronize()
    finally:
        lib.st2_lstm_coop_set_exchange(2)
    assert ops.lstm_coop_status() == 0 and ops.status() == 0
    assert torch.equal(out, ref)


@pytest.mark.parametrize("mode", ["coop", "single"])
@pytest.mark.parametrize("B,N,ragged", [(2, 17, False), (3, 40, True), (1, 33, False), (9, 25, True), (32, 60, False),
                                        (13, 101, True)])
def test_lstm_bidir_matches_torch_lstm(B, N, ragged, mode, monkeypatch):
    """Input projection on st2_conv1d + the recurrence kernel == nn.LSTM(bidirectional) with pack/pad semantics, for
    both recurrence kernels: `coop` = st2_lstm_bidir_coop (register-resident W_hh over 8 CUs per group; utterance
    blocks of 1 / 4 / 8, partial blocks, ragged lengths inside a block), `single` = st2_lstm_bidir."""
    from styletts2_amd.text import EngineLSTM
    monkeypatch.setattr(_hooks, "lstm", mode)
    torch.manual_seed(N)
    lstm = EngineLSTM(640, 256)
    ref_lstm = torch.nn.LSTM(640, 256, 1, batch_first=True, bidirectional=True)
    ref_lstm.load_state_dict(lstm.state_dict())
    x = torch.randn(B, N, 640)
    if ragged:
        lengths = torch.randint(1, N + 1, (B,), generator=torch.Generator().manual_seed(B))
        lengths[0] = N
        if B > 2:
            lengths[2] = 5
    else:
        lengths = torch.full((B,), N)
    with torch.no_grad():
        packed = torch.nn.utils.rnn.pack_padded_sequence(x, lengths, batch_first=True, enforce_sorted=False)
        ref, _ = torch.nn.utils.rnn.pad_packed_sequence(ref_lstm(packed)[0], batch_first=True, total_length=N)
    lstm = lstm.to(DEV)
    lens = lengths.to(torch.int32).to(DEV) if ragged else None
    out = lstm.forward_cm(g(x).transpose(1, 2).contiguous(), lens).transpose(1, 2)
    torch.cuda.synchronize()
    if mode == "coop":
        assert ops.lstm_coop_status() == 0, "a cooperative LSTM group timed out"
    assert out.shape == ref.shape
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    if not ragged:
        y, _ = lstm(g(x))
        assert torch.equal(y, out)  # bitwise reproducible


XS_VARIANT_CASES = [
    # (B, C, L, ks, dil, aligned rows): C_out = 256 -> 2 row blocks (the XCD-aware order applies), grids divisible by 8 or not
    (8, 256, 1000, 7, 1, True),
    (4, 256, 1531, 11, 5, True),    # odd length: edge tiles, the generic epilogue
    (8, 256, 2000, 3, 3, True),
    (3, 256, 777, 7, 3, False),     # unaligned dense tensors (4-byte epilogue), grid not divisible by 8 (swizzle refused)
    (2, 512, 640, 3, 1, True),      # 4 row blocks
    (4, 1090, 400, 3, 1, True),     # ragged C_in (zero-padded last chunk), 2 row blocks
    (6, 128, 3001, 11, 1, True),    # one row block: the swizzle bit is a no-op
]


@pytest.mark.parametrize("B,C,L,ks,dil,aligned", XS_VARIANT_CASES)
def test_xs_variants_are_bitwise_identical(B, C, L, ks, dil, aligned):
    """Every build the autotuner may pick (include/st2.h st2_conv_tune: tile shape / occupancy, XCD-aware tile order)
    issues the same products in the same order through the same epilogue: output AND InstanceNorm statistics are
    bit-identical to the rule's build -- the choice is a matter of time only."""
    gen = torch.Generator().manual_seed(4242 + ks)
    C_out = 256 if C == 1090 else C
    pitch = (L + 31) // 32 * 32 if aligned else L
    x = torch.randn(B, C, L, generator=gen) * 1.5
    w = torch.randn(C_out, C, ks, generator=gen) / math.sqrt(C * ks)
    wt = weights.pack_conv_f16s(w).to(DEV)
    bias = g(torch.randn(C_out, generator=gen))
    res = torch.empty(B, C_out, pitch, device=DEV)[:, :, :L].copy_(torch.randn(B, C_out, L, generator=gen))
    xs = ops.activate(g(x))
    pad = (ks - 1) * dil // 2
    outs = {}
    variants = [-1, 0, 2] + ([1, 3] if ks >= 7 or ks == 3 else [])
    try:
        for v in variants:
            ops.conv_tune_set(ks, C, C_out, L, B, v)
            out = torch.full((B, C_out, pitch), float("nan"), device=DEV)[:, :, :L]
            y, st = ops.conv1d_xs(xs, wt, C_out, ks, dil=dil, pad_left=pad, bias=bias, res=res, out=out, want_stats=True)
            torch.cuda.synchronize()
            outs[v] = (y.clone(), st.clone())
    finally:
        ops.conv_tune_set(ks, C, C_out, L, B, -1)
    ref = R.conv1d(x, weights.pack_conv_f16s(w), C_out, ks, dil=dil, pad_left=pad, bias=bias.cpu(), res=res.cpu(), pro=R.PRO_NONE)
    assert rel_err(outs[-1][0], ref) < 3e-6
    for v in variants[1:]:
        assert torch.equal(outs[v][0], outs[-1][0]), "variant %d differs from the rule's build" % v
        assert torch.equal(outs[v][1], outs[-1][1]), "variant %d: statistics differ" % v
    assert ops.status() == 0


def test_conv_autotune_measures_keeps_results_and_survives_aliasing():
    """st2_conv_tune(1): the first launch of a class times its candidate builds into a SCRATCH output -- the caller's
    tensors are only read, so a conv that accumulates into its own output (res2 aliases y: the MRF sum of
    Modules/istftnet.py:366-373) is applied exactly once -- records them, and later launches run the winner with unchanged
    results."""
    B, C, L, ks, dil = 8, 256, 4000, 7, 3  # 29 GFLOP: above the tuner's 20 GFLOP floor
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(B, C, L, generator=gen)
    w = torch.randn(C, C, ks, generator=gen) / math.sqrt(C * ks)
    wt = weights.pack_conv_f16s(w).to(DEV)
    acc0 = torch.randn(B, C, L, generator=gen)
    xs = ops.activate(g(x))
    pad = (ks - 1) * dil // 2

    def run():
        acc = g(acc0).clone()
        ops.conv1d_xs(xs, wt, C, ks, dil=dil, pad_left=pad, res2=acc, out=acc)  # acc += conv(x)
        torch.cuda.synchronize()
        return acc
    ops.conv_tune_set(ks, C, C, L, B, -1)
    plain = run()
    with ops.conv_autotune(reset=True):
        tuned_first = run()
        tuned_again = run()
    after = run()
    table = [r for r in ops.conv_tune_table() if (r["ks"], r["C_in"], r["L"], r["B"]) == (ks, C, L, B)]
    try:
        assert len(table) == 1 and len(table[0]["candidates"]) >= 3, table
        assert all(c["ms"] > 0 for c in table[0]["candidates"]), table
        assert table[0]["chosen"] in [c["variant"] for c in table[0]["candidates"]]
        for t in (tuned_first, tuned_again, after):
            assert torch.equal(t, plain)
    finally:
        from styletts2_amd import _lib
        _lib.load().st2_conv_tune(-1)
    assert ops.conv_tune_table() == []


def test_probe_box_reports_a_plausible_mi355x():
    """st2_probe_box: the micro-probe bench.py embeds in its JSON line.  Sanity bounds only (this is a measurement)."""
    p = ops.probe_box(0)
    assert p["cus"] >= 64 and p["census"]["cus_seen"] <= p["cus"] and p["census"]["workgroups"] == 2048
    assert p["mfma"]["random"]["tflops"] > 100 and 0.5 < p["mfma"]["random"]["clock_ghz"] < 3.0
    assert p["mfma"]["zero"]["tflops"] >= p["mfma"]["random"]["tflops"] * 0.9
    assert len(p["sets"]) == 5 and all(s["chase_ns_median"] > 50 and s["stream8_gbps"] > 100 for s in p["sets"])
    assert p["sets"][0]["chase_ns_median"] < p["sets"][-1]["chase_ns_median"] * 1.2  # a cache level is not slower than HBM
    assert p["hbm_copy"]["gbps_read_plus_write"] > 500


def test_cu_health_probe_and_masked_streams():
    """st2_probe_cu_health runs an instrumented copy of the conv kernel and returns the CU mask of the device without its
    degraded CUs (none on a healthy box); a stream made from that mask computes what the default stream computes."""
    from styletts2_amd import pipeline
    rep, mask, n = ops.probe_cu_health()
    cus = rep["cus"]
    assert rep["workgroups"] == 2048 and rep["epilogue_cycles_median"] > 1000 and len(rep["xcd_end_us"]) == 8
    assert 0 <= n <= cus // 8 and n <= rep["n_slow_cus"]
    assert sum(bin(w).count("1") for w in mask) == cus - n
    ms = pipeline.MaskedStreams(DEV, mask)
    try:
        gen = torch.Generator().manual_seed(3)
        x = g(torch.randn(4, 256, 2000, generator=gen))
        w = weights.pack_conv_f16s(torch.randn(256, 256, 7, generator=gen) / math.sqrt(256 * 7)).to(DEV)
        xs = ops.activate(x)
        ref = ops.conv1d_xs(xs, w, 256, 7, pad_left=3)
        torch.cuda.synchronize()
        ms.main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ms.main):
            xs2 = ops.activate(x)
            out = ops.conv1d_xs(xs2, w, 256, 7, pad_left=3)
        ms.main.synchronize()
        assert torch.equal(out, ref)
    finally:
        ms.close()
