"""Per-layer operand scales of the split-f16 convs (include/st2.h st2_calibrate / st2_calibration_*), the parts that need no
GPU: the scale formula, the site table of an engine (one site per packed conv weight, named by its state_dict key), and that a
written table reaches every conv launch of the C++ plans (CPU backend: the contracts honour whatever power of two they are
handed, so the plan output must not move beyond the contract tolerance).  The measured side -- st2_debug_headroom records
folded by st2_calibrate on the device -- is tests/test_calibration_gpu.py."""
import math

import pytest
import torch

import _cpu_backend as CB
from _cpu_backend import cpu_backend
from _util import decoder_kwargs, manifest
from oracle import st2_oracle as O
from styletts2_amd import _lib, engine
from styletts2_amd.decoder import Decoder
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)


def test_scale_formula():
    lib = _lib.load()
    for margin in (0, 3, 5):
        for v in (1e-6, 3.1e-4, 0.02, 0.5, 1.0, 7.9, 8.0, 8.1, 250.0, 9000.0, 65504.0, 4e6):
            s = lib.st2_calibration_scale(v, margin)
            m, _ = math.frexp(s)
            assert m == 0.5, "power of two"
            top = 2.0 ** (16 - margin)
            assert top / 2 <= v * s < top, (v, s, margin)  # the largest scaled operand sits in the top octave below 2^(16-m)
    assert lib.st2_calibration_scale(0.0, 3) == 0.0 and lib.st2_calibration_scale(-1.0, 3) == 0.0
    assert lib.st2_calibration_scale(float("nan"), 3) == 0.0 and lib.st2_calibration_scale(float("inf"), 3) == 0.0


@pytest.fixture
def small_decoder():
    dc = manifest("ljspeech")["config"]["decoder"]
    dec = Decoder(**decoder_kwargs(dc)).eval()
    synth.init_synthetic_(dec, 1)
    return dc, dec


def test_site_table_and_written_scales_reach_every_launch(small_decoder):
    dc, dec = small_decoder
    asr, F0, N, s, noise = synth.decoder_inputs(2, 6, 3)
    to = {}
    with torch.no_grad():
        ref = O.decoder(dec.state_dict(), dc, asr, F0, N, s, noise=noise, taps=to)
    with cpu_backend():
        eng = engine.build_decoder_engine(dec, None)
        sites = eng.calibration()
        names = [r["name"] for r in sites]
        # one site per split-f16 packed weight, named by the state_dict key it came from
        assert len(sites) == len(set(names)) > 60
        for want in ("decoder.asr_res.0.weight", "decoder.encode.conv1.weight", "decoder.generator.ups.0.weight",
                     "decoder.generator.resblocks.0.convs1.0.weight", "decoder.generator.conv_post.weight"):
            assert want in names, want
        assert all(r["x_scale"] == 0.0 and r["seen"] == 0.0 for r in sites), "a fresh engine runs by rule"
        gen0 = eng.calib_gen

        CB.X_SCALES_SEEN.clear()
        base = eng.decoder_forward(asr, F0, N, s, noise=noise, har=to["har"])
        by_rule = list(CB.X_SCALES_SEEN)
        assert by_rule and {x for _, x in by_rule} <= {1.0, 8.0}

        # a table of distinct powers of two: every launch must carry its site's value, none the rule's
        table = [2.0 ** (3 + i % 7) for i in range(len(sites))]
        eng.set_calibration(table)
        assert eng.calib_gen == gen0 + 1
        assert eng.calibration_scales() == table
        CB.X_SCALES_SEEN.clear()
        CB.EXPECT_RULE[0] = False
        try:
            out = eng.decoder_forward(asr, F0, N, s, noise=noise, har=to["har"])
        finally:
            CB.EXPECT_RULE[0] = True
        seen = [x for _, x in CB.X_SCALES_SEEN]
        assert len(seen) == len(by_rule)
        assert set(seen) <= set(table) and len(set(seen)) == 7
        # powers of two rescale exactly: the plan's output stays at the oracle (contract tolerance), whatever the table
        assert (out - ref).pow(2).mean().sqrt().item() < 1e-5
        assert (out - base).abs().max().item() < 1e-5

        # clearing restores the rule bit for bit
        eng.set_calibration(None)
        CB.X_SCALES_SEEN.clear()
        again = eng.decoder_forward(asr, F0, N, s, noise=noise, har=to["har"])
        assert CB.X_SCALES_SEEN == by_rule and torch.equal(again, base)


def test_calibration_write_validates_and_is_dropped_by_a_reload(small_decoder):
    dc, dec = small_decoder
    with cpu_backend():
        eng = engine.build_decoder_engine(dec, None)
        n = len(eng.calibration())
        with pytest.raises(_lib.St2Error, match="conv sites"):
            eng.set_calibration([1.0] * (n - 1))
        with pytest.raises(_lib.St2Error, match="power of two"):
            eng.set_calibration([3.0] + [1.0] * (n - 1))
        with pytest.raises(_lib.St2Error, match="power of two"):
            eng.set_calibration([-2.0] + [1.0] * (n - 1))
        table = [0.0 if i % 5 == 0 else 2.0 ** (i % 4) for i in range(n)]  # 0 = this site by rule
        eng.set_calibration(table)
        assert eng.calibration_scales() == table
        # new weights, no table: scales belong to the checkpoint they were measured on (another checkpoint of the SAME architecture
        # would run clamped or with subnormal lo halves, silently) -- the caller re-calibrates or writes the table saved beside it
        synth.init_synthetic_(dec, 2)
        eng.load_module("decoder.", dec)
        eng.finalize(1, None)
        assert eng.calibration_scales() == [0.0] * n
        # ... an engine holding another layout refuses it by size
        lib = dict(manifest("libritts")["config"]["decoder"])
        dec2 = Decoder(**decoder_kwargs(lib)).eval()
        synth.init_synthetic_(dec2, 1)
        eng2 = engine.build_decoder_engine(dec2, None)
        assert len(eng2.calibration()) != n
        with pytest.raises(_lib.St2Error, match="conv sites"):
            eng2.set_calibration(table)


def test_replacing_a_calibrated_engine_warns_and_device_spellings_agree(small_decoder):
    """Advisor, round 5: `eng.device != dev` compared the caller's spelling ('cuda:0' vs torch.device('cuda')), rebuilt the engine
    and lost its table without a word.  Devices are normalised before the comparison, and a calibrated engine never goes silently."""
    import warnings
    dc, dec = small_decoder
    assert engine.norm_device(None) is None
    assert engine.norm_device("cuda:1") == torch.device("cuda", 1) == engine.norm_device(torch.device("cuda:1"))
    with cpu_backend():
        eng = engine.build_decoder_engine(dec, None)
        assert engine.same_device(eng, None) and not engine.same_device(None, None)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            engine.replaced(eng, "decoder")  # by rule: nothing to lose, nothing said
        n = len(eng.calibration())
        eng.set_calibration([2.0] * n)
        with pytest.warns(RuntimeWarning, match="operand scales are\\s+dropped"):
            engine.replaced(eng, "decoder")
        dec._eng = eng
        with pytest.warns(RuntimeWarning, match="decoder engine is being rebuilt"):
            dec._pk = None  # what load_state_dict / .to() do
        assert dec._eng is None
