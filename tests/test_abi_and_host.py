"""C-ABI surface (no compute: there is no GPU here) and host-side logic."""
import ctypes
import math
import os
import re

import pytest
import torch

from styletts2_amd import _lib, ops, weights
from styletts2_amd.utils import length_to_mask, recursive_munch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "st2.h")).read()
    declared = set(re.findall(r"\b(st2_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"st2_conv_desc"}
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), "libst2_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.st2_abi_version() == _lib.ABI_VERSION
    assert lib.st2_sizeof_conv_desc() == ctypes.sizeof(_lib.ConvDesc)


def test_no_cpu_path():
    with pytest.raises(_lib.St2Error):
        ops.instnorm_stats(torch.zeros(1, 2, 3))
    with pytest.raises(_lib.St2Error):
        ops.conv1d(torch.zeros(1, 2, 8), torch.zeros(6, 4), 4, 3)


def test_weight_norm_fold_matches_torch():
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 5, 3))
    conv.weight_g.data.mul_(1.7)
    w = weights.fold_weight_norm(conv.weight_g.data, conv.weight_v.data)
    x = torch.randn(2, 6, 9)
    assert torch.allclose(torch.nn.functional.conv1d(x, w, conv.bias), conv(x), atol=1e-6)
    ct = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 4, 2, padding=1))  # dim 0 = C_in
    assert ct.weight_g.shape == (6, 1, 1)
    w = weights.fold_weight_norm(ct.weight_g.data, ct.weight_v.data)
    assert torch.allclose(torch.nn.functional.conv_transpose1d(x, w, ct.bias, stride=2, padding=1), ct(x), atol=1e-6)


def test_pack_conv_layout():
    w = torch.arange(5 * 3 * 2, dtype=torch.float32).reshape(5, 3, 2)
    wt = weights.pack_conv(w)
    assert wt.shape == (6, 8) and wt[:, 5:].abs().sum() == 0
    for co in range(5):
        for ci in range(3):
            for t in range(2):
                assert wt[ci * 2 + t, co] == w[co, ci, t]


def test_expand_by_durations_equals_one_hot_matmul():
    """The contract of st2_expand_by_durations (oracle/ops_ref.py) is the notebooks' one-hot alignment matmul, bit for
    bit (the matmul only ever adds zeros); the HIP kernel is held to the same contract in tests/test_ops_gpu.py."""
    from oracle.ops_ref import expand_by_durations
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 6, generator=g)
    dur = torch.tensor([[1, 2, 3, 1, 2, 3], [2, 2, 2, 2, 2, 2]])
    T = 12
    aln = torch.zeros(2, 6, T)
    for b in range(2):
        c = 0
        for i in range(6):
            aln[b, i, c:c + int(dur[b, i])] = 1
            c += int(dur[b, i])
    assert torch.equal(expand_by_durations(x, dur, T), x @ aln)


def test_utils():
    m = length_to_mask(torch.tensor([3, 1]))
    assert m.tolist() == [[False, False, False], [False, True, True]]
    cfg = recursive_munch({"a": {"b": [1, {"c": 2}]}})
    assert cfg.a.b[1].c == 2


def test_plbert_has_no_cpu_fallback():
    """The engine PL-BERT must not silently run the HF forward on CPU tensors (there is none behind it) nor accept HF-only options."""
    import torch
    from _util import manifest
    from styletts2_amd import models
    from styletts2_amd._lib import St2Error
    bert = models.load_plbert(manifest("ljspeech")["plbert"]).eval()
    ids = torch.zeros(1, 5, dtype=torch.long)
    with pytest.raises(St2Error):
        bert(ids, attention_mask=torch.ones(1, 5, dtype=torch.int32))
    with pytest.raises(TypeError):
        bert(ids, attention_mask=None, output_hidden_states=True)


def test_xs_conv_hot_builds_do_not_spill(tmp_path):
    """Every build of the dominant conv family keeps its k loop in registers (no scratch_ instruction between the first and
    the last v_mfma of the gfx950 code objects), the 3-workgroups-per-CU builds (conv1d_xs_kernel_o3) within 168 VGPRs, and what
    little scratch a build uses outside the loop (rare epilogue modes) stays <= 64 bytes."""
    import shutil
    import subprocess
    tools = "/opt/rocm/lib/llvm/bin"
    objdump, readelf = os.path.join(tools, "llvm-objdump"), os.path.join(tools, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm LLVM binutils not installed")
    objs = [os.path.join(ROOT, "styletts2_amd", "csrc", "build", "st2_conv1d_xs_k%d.o" % i) for i in range(4)]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("objects not built (run __graft_entry__.build())")
    seen = 0
    for o in objs:
        local = shutil.copy(o, tmp_path)
        subprocess.check_call([objdump, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp_path) if f.startswith(os.path.basename(o)) and "gfx950" in f]
        assert co, "no gfx950 code object in %s" % o
        co = os.path.join(tmp_path, co[0])
        notes = subprocess.check_output([readelf, "--notes", co], text=True)
        kernels = re.findall(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)(?:.|\n)*?\.vgpr_count:\s+(\d+)"
                             r"\n\s+\.vgpr_spill_count:\s+(\d+)", notes)
        for name, priv, vgpr, spill in kernels:
            if "conv1d_xs_kernel" not in name:
                continue
            # No scratch and no spill in any build on the hot path (advisor, round 4: the allowance of round 4 was wider than
            # the regression it accommodated).  The one exception is named: the 64-row x 256-column build with 16-channel
            # chunks at 3 workgroups / CU (<KS, 16, 2, 2, 4>: C_out in (32, 64] at k >= 3 on the xs path -- no layer of the
            # five BASELINE configurations routes there, prefer_fused takes C <= 64) parks one quad (<= 64 B) in scratch in
            # its residual + MRF-accumulator epilogue modes, outside the k loop (checked on the disassembly below).
            rare = re.search(r"conv1d_xs_kernel_o3ILi\d+ELi16ELi2ELi2ELi4EE", name) is not None
            if rare:
                assert int(priv) <= 64 and int(spill) <= 16, "%s: %s B of scratch, %s spills" % (name, priv, spill)
            else:
                assert int(priv) == 0 and int(spill) == 0, "%s: %s B of scratch, %s VGPR spills" % (name, priv, spill)
            if "conv1d_xs_kernel_o3" in name:
                seen += "conv1d_xs_kernel_o3" in name
                assert int(vgpr) <= 168, "%s uses %s VGPRs: 2 workgroups per CU, not 3" % (name, vgpr)
            else:
                assert int(vgpr) <= 256, "%s uses %s VGPRs" % (name, vgpr)
        dis = subprocess.check_output([objdump, "-d", co], text=True)
        cur, body = None, []

        def check(cur, body):
            if cur is None or "conv1d_xs_kernel" not in cur:
                return
            mf = [i for i, l in enumerate(body) if "v_mfma" in l]
            sc = [i for i, l in enumerate(body) if "scratch_" in l]
            if mf:
                assert not [i for i in sc if mf[0] < i < mf[-1]], "%s: scratch instruction inside the k loop" % cur
        for line in dis.splitlines():
            if line.endswith(">:"):
                check(cur, body)
                cur, body = line, []
            else:
                body.append(line)
        check(cur, body)
        os.remove(co)
    assert seen >= 9, "expected the o3 builds of every kernel size, saw %d" % seen


def test_splitk_workspace_query_is_a_function_of_the_geometry():
    """st2_conv1d_f16s_splitk_bytes (host-only): skinny launches -- < 128 workgroups and >= 8 K chunks -- are split into
    up to 8 slices (~256 workgroups in total), everything else is not; both plans size their workspace with this call."""
    import ctypes as C
    lib = _lib.load()

    def q(B, C_in, C_out, L, ks):
        d = _lib.ConvDesc()
        d.B, d.C_in, d.C_out, d.L_in, d.L_out, d.ks, d.dil = B, C_in, C_out, L, L, ks, 1
        return lib.st2_conv1d_f16s_splitk_bytes(C.byref(d))

    # slices are stored in the accumulator layout: whole 128 x 128 tiles (C_out > 64)
    assert q(1, 1024, 2048, 100, 1) == 8 * 1 * 2048 * 128 * 4        # 16 workgroups, 32 chunks -> 8 slices
    assert q(1, 2048, 1024, 112, 1) == 8 * 1 * 1024 * 128 * 4
    assert q(4, 1024, 1024, 100, 1) == 8 * 4 * 1024 * 128 * 4        # 32 workgroups -> 256 / 32 = 8 slices
    assert q(8, 1024, 1024, 100, 1) == 4 * 8 * 1024 * 128 * 4        # 64 workgroups -> 4 slices
    assert q(32, 1024, 1024, 100, 1) == 0                            # 256 workgroups: a split loses (measured)
    assert q(1, 128, 1024, 100, 1) == 0                              # 4 chunks: nothing to split
    assert q(1, 512, 512, 37, 5) == 8 * 512 * 128 * 4                # k = 5: 16-channel chunks
    assert q(1, 512, 40, 37, 5) == 8 * 64 * 256 * 4                  # 64 x 256 tiles for 32 < C_out <= 64
    assert q(0, 0, 0, 0, 1) == 0


def test_conv_routing_rule_is_the_same_in_both_plans():
    """ops.prefer_fused (Python plans) and conv() in csrc/st2_engine.hip (C++ plans) must route a layer to the same
    kernel, or the plans stop being bitwise equal: the constants are compared at the source level."""
    from styletts2_amd import ops
    src = open(os.path.join(ROOT, "styletts2_amd", "csrc", "st2_engine.hip")).read()
    m = re.search(r"FUSED_MAX_C = (\d+), FUSED_K3_MAX_C = (\d+)", src)
    assert m and (int(m.group(1)), int(m.group(2))) == (ops.FUSED_MAX_C, ops.FUSED_K3_MAX_C)
    m = re.search(r"XS_MIN_L = (\d+)", src)
    assert m and int(m.group(1)) == ops.XS_MIN_L
    m = re.search(r"XS_MIN_C_PLAIN = (\d+)", src)
    assert m and int(m.group(1)) == ops.XS_MIN_C_PLAIN
    assert ops.prefer_fused(ops.PRO_ADAIN_SNAKE, 64, 11) and ops.prefer_fused(ops.PRO_ADAIN_SNAKE, 128, 3)
    assert not ops.prefer_fused(ops.PRO_ADAIN_SNAKE, 128, 7) and not ops.prefer_fused(ops.PRO_NONE, 32, 3)
