"""Co-residency canaries: product kernels next to ANOTHER queue's matrix-pipe work must return the idle result bit for bit.

Collected LAST on purpose (file name): these are stress tests of the box as much as of the code, and `pytest -x` must not lose the
oracle-parity files behind them (GPUTEST_r05 stopped here with 58 tests unrun).

Background (tools/simd_hazard_repro.hip): round 5 saw the BiLSTM kernels return other bits while narrow-tile
convs ran on a second stream.  Round 6 traced it to gfx950 itself: a packed-f32 VALU op whose op_sel takes the HIGH dword of src1 for
the low result lane (`v_pk_fma_f32 ... op_sel:[0,1,0]`, which hipcc's SLP vectorizer emitted in exactly the two BiLSTM kernels) returns
a wrong low half in lanes 48-63 while another wave of the CU issues MFMAs in certain cadences.  The library no longer contains the
encoding (tools/check_isa.py gates the build; tests/test_isa_gate.py); what is checked here is the behaviour: every kernel class of
the front, and the whole path, next to (a) the library's own small-grid convs and (b) `st2_probe_mfma_stream`, which issues the MFMA
cadences that hit the old kernels in 90-100 % of their calls."""
import pytest
import torch

from styletts2_amd import _hooks, ops, weights

pytestmark = pytest.mark.gpu
DEV = "cuda"
KINDS = (0, 1, 2)  # st2_probe_mfma_stream: 16x16x32 chain, 32x32x16 isolated groups of three, 16x16x32 four chains


def g(t):
    return t.to(DEV)


def _next_to(load, fn, n_calls, what):
    """`fn()` n_calls times on a side stream while `load()` has filled the current stream; every result == the idle one."""
    ref = fn()
    ref = [r.clone() for r in (ref if isinstance(ref, (tuple, list)) else (ref,))]
    torch.cuda.synchronize()
    side = ops.aux_stream(torch.device(DEV, torch.cuda.current_device()), 0, index=3)
    outs = []
    side.wait_stream(torch.cuda.current_stream())
    load()
    with torch.cuda.stream(side):
        for _ in range(n_calls):
            o = fn()
            outs.append(o if isinstance(o, (tuple, list)) else (o,))
    torch.cuda.synchronize()
    bad = sum(any(not torch.equal(a, b) for a, b in zip(o, ref)) for o in outs)
    assert bad == 0, "%d of %d %s calls differ from the idle run" % (bad, n_calls, what)


def _mfma(kind, launches):
    return lambda: [ops.mfma_load(kind) for _ in range(launches)]


@pytest.mark.parametrize("mode", ["coop", "single"])
def test_lstm_is_reproducible_next_to_one_utterance_convs_on_another_stream(mode):
    """The round-5 canary, unchanged in substance: k = 3 convs in 32-column tiles (what one utterance launches), k = 7 in 128-column
    tiles, on one stream; the BiLSTM of another sentence on a second.  Failed 3 / 30 on the driver's box in round 5."""
    gen = torch.Generator().manual_seed(0)
    lx = ops.activate(g(torch.randn(1, 256, 5680, generator=gen)))
    w3 = weights.pack_conv_f16s(torch.randn(256, 256, 3, generator=gen) / 30).to(DEV)
    w7 = weights.pack_conv_f16s(torch.randn(256, 256, 7, generator=gen) / 40).to(DEV)
    y = torch.empty(1, 256, 5680, device=DEV)
    G = g(torch.randn(1, 2048, 24, generator=gen))
    whh = g(torch.randn(2, 256, 1024, generator=gen) / 16).contiguous()
    with _hooks.override(lstm=mode):
        for w, ks, n in ((w3, 3, 300), (w7, 7, 150)):
            _next_to(lambda: [ops.conv1d_xs(lx, w, 256, ks, pad_left=(ks - 1) // 2, out=y, want_stats=True) for _ in range(n)],
                     lambda: ops.lstm_bidir(G, whh), 40 if mode == "coop" else 12, "BiLSTM (next to k = %d convs)" % ks)
    assert ops.status(clear=True) == 0


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mode,B,N", [("coop", 1, 24), ("single", 1, 24), ("coop", 32, 100), ("single", 3, 60)])
def test_lstm_next_to_the_mfma_cadences_that_hit_round5(kind, mode, B, N):
    """Both recurrences, one utterance and the two-stream batch shape, next to the provoking MFMA streams: round 5's kernels differed
    in 90-100 % of such calls (profiles/r06a_repro.log: 600 / 600), these must not differ at all."""
    gen = torch.Generator().manual_seed(B * 1000 + N)
    G = g(torch.randn(B, 2048, N, generator=gen))
    whh = g(torch.randn(2, 256, 1024, generator=gen) / 16).contiguous()
    with _hooks.override(lstm=mode):
        _next_to(_mfma(kind, 40 if B == 1 else 120), lambda: ops.lstm_bidir(G, whh), 30 if B == 1 else 10,
                 "BiLSTM %s B=%d (MFMA cadence %d)" % (mode, B, kind))
    assert ops.status(clear=True) == 0


def _front_kernels():
    """One representative call per kernel class the front / prosody plans launch besides the BiLSTM."""
    gen = torch.Generator().manual_seed(11)
    r = lambda *s: g(torch.randn(*s, generator=gen))
    cases = {}
    q, k, v = r(4, 512, 100), r(4, 512, 100), r(4, 512, 100)
    cases["attention"] = lambda: ops.attention(q, k, v, 8, 0.125)
    s, wt, b = r(32, 128), r(128, 2048), r(2048)
    cases["style_fc"] = lambda: ops.style_fc(s, wt, b)
    x = r(4, 768, 100)
    st = ops.colnorm_stats(x)
    gam, bet = r(1, 768), r(1, 768)
    cases["colnorm_stats"] = lambda: ops.colnorm_stats(x)
    cases["colnorm_apply"] = lambda: ops.colnorm_apply(x, st, gam, bet)
    xa = r(2, 512, 800)
    sta = ops.instnorm_stats(xa)
    ga, ba, al = r(2, 512), r(2, 512), g(torch.rand(512, generator=gen) + 0.5)
    cases["instnorm_stats"] = lambda: ops.instnorm_stats(xa)
    cases["act_split adain+snake"] = lambda: ops.activate(xa, pro=ops.PRO_ADAIN_SNAKE, stats=sta, gamma=ga, beta=ba, alpha=al).data
    w3 = weights.pack_conv_f16s(torch.randn(512, 512, 3, generator=gen) / 40).to(DEV)
    lx = ops.activate(xa, pro=ops.PRO_ADAIN_SNAKE, stats=sta, gamma=ga, beta=ba, alpha=al)
    cases["conv1d_xs k3 (+ statistics)"] = lambda: ops.conv1d_xs(lx, w3, 512, 3, pad_left=1, want_stats=True)
    xf = r(2, 64, 4000)
    wf = weights.pack_conv_f16s(torch.randn(64, 64, 7, generator=gen) / 20).to(DEV)
    stf = ops.instnorm_stats(xf)
    gf, bf, af = r(2, 64), r(2, 64), g(torch.rand(64, generator=gen) + 0.5)
    cases["conv1d fused k7 C64"] = lambda: ops.conv1d(xf, wf, 64, 7, pad_left=3, pro=ops.PRO_ADAIN_SNAKE, stats=stf, gamma=gf, beta=bf,
                                                       alpha=af)
    xt = r(1, 1024, 400)
    wl = weights.pack_conv_f16s(torch.randn(2048, 1024, 1, generator=gen) / 32).to(DEV)
    cases["token GEMM k1 1024->2048 (GELU)"] = lambda: ops.conv1d(xt, wl, 2048, 1, act=ops.ACT_GELU)
    return cases


@pytest.mark.parametrize("kind", KINDS)
def test_front_kernel_classes_next_to_the_mfma_cadences(kind):
    for name, fn in _front_kernels().items():
        _next_to(_mfma(kind, 30), fn, 12, name + " (MFMA cadence %d)" % kind)
    assert ops.status(clear=True) == 0


def test_whole_path_next_to_the_mfma_cadences_and_small_grid_convs():
    """tokens -> waveform for one short utterance (the C++ plans: every kernel of the path, BiLSTMs included) while the second queue runs
    the provoking MFMA streams and the 32-column conv build: bitwise the idle run, five times per load."""
    from benchdata import manifest, synth
    from styletts2_amd import models, pipeline
    import bench
    man = manifest("ljspeech")
    model = bench.build(man)
    for i, k in enumerate(bench.KEYS):
        synth.init_synthetic_(model[k], 10 + i)
        model[k].eval().to(DEV)
    sampler = models.make_sampler(model)
    gen = torch.Generator().manual_seed(5)
    N = 40
    tokens = torch.randint(1, 170, (1, N), generator=gen).to(DEV)
    lengths = torch.tensor([N])
    noise = g(torch.randn(1, 1, 256, generator=gen))
    durations = torch.full((1, N), 3, dtype=torch.long).to(DEV)
    # what a call otherwise draws itself (ADPM2 ancestral noise, SineGen noise) is pinned: the comparison is bit for bit
    step_noise = g(torch.randn(4, 1, 1, 256, generator=gen))
    sine_noise = g(torch.randn(1, 3 * N * 600, 9, generator=gen))

    def run():
        return pipeline.inference(model, sampler, tokens, lengths, noise, diffusion_steps=5, embedding_scale=1.0, durations=durations,
                                  total_frames=3 * N, step_noise=step_noise, sine_noise=sine_noise)
    lx = ops.activate(g(torch.randn(1, 256, 5680, generator=gen)))
    w3 = weights.pack_conv_f16s(torch.randn(256, 256, 3, generator=gen) / 30).to(DEV)
    y = torch.empty(1, 256, 5680, device=DEV)
    loads = [_mfma(k, 400) for k in KINDS]
    loads.append(lambda: [ops.conv1d_xs(lx, w3, 256, 3, pad_left=1, out=y, want_stats=True) for _ in range(1500)])
    for i, load in enumerate(loads):
        _next_to(load, run, 5, "tokens -> waveform (load %d)" % i)
    ops.check_status()
