"""Build gate against the gfx950 packed-f32 `op_sel` hazard (tools/check_isa.py): the linked library holds no
v_pk_{fma,mul,add}_f32 with an op_sel bit, and the checker itself recognises the encodings round 5's BiLSTM kernels contained."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location("st2_check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_flags_the_round5_encodings_and_nothing_else():
    C = _checker()
    kernels = {
        "lstm_r05": ["v_pk_fma_f32 v[170:171], v[8:9], v[154:155], v[170:171] op_sel:[0,1,0]",       # the one that is wrong on gfx950
                     "v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]",
                     "v_pk_fma_f32 v[158:159], v[10:11], v[138:139], v[158:159] op_sel_hi:[1,0,1]",  # low-dword broadcast: measured clean
                     "v_pk_fma_f32 v[44:45], v[2:3], v[4:5], v[44:45]",
                     "v_pk_mul_f32 v[0:1], s[2:3], v[4:5] op_sel_hi:[0,1]",
                     "v_pk_add_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]",
                     "v_fma_f32 v1, v2, v3, v1", "v_mfma_f32_32x32x16_f16 v[0:15], v[54:57], v[28:31], v[0:15]"],
        "clean": ["v_pk_fma_f32 v[2:3], v[2:3], s[0:1], v[6:7]", "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]"],
    }
    bad = C.violations(kernels)
    assert [(k, i.split()[0]) for k, i in bad] == [("lstm_r05", "v_pk_fma_f32"), ("lstm_r05", "v_pk_add_f32")]


def test_library_passes_the_isa_gate():
    C = _checker()
    lib = os.path.join(ROOT, "styletts2_amd", "libst2_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built (python -m styletts2_amd._build)")
    kernels = C.disassemble(lib)
    if kernels is None:
        pytest.skip("llvm-objdump not found")
    bad = C.violations(kernels)
    assert not bad, "packed-f32 ops with op_sel in the library (python tools/check_isa.py lists them): %r" % (bad[:5],)
    lstm = [k for k in kernels if "lstm_coop_kernel" in k or "lstm_recurrence_kernel" in k]
    assert len(lstm) >= 5  # both recurrences are in the library that was checked
    # the cooperative kernel still runs its mat-vec packed (plain encoding): the fix did not fall back to scalar code
    coop4 = [k for k in lstm if "lstm_coop_kernelILi4ELi2E" in k]
    assert coop4 and sum("v_pk_fma_f32" in i for i in kernels[coop4[0]]) >= 256


def test_build_rejects_a_library_that_fails_the_gate():
    src = open(os.path.join(ROOT, "styletts2_amd", "_build.py")).read()
    assert "isa_gate(verbose)" in src and '"st2_lstm.hip": ["-fno-slp-vectorize"]' in src
