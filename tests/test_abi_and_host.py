"""C-ABI surface (no compute: there is no GPU here) and host-side logic."""
import ctypes
import math
import os
import re

import pytest
import torch

from styletts2_amd import _lib, ops, weights
from styletts2_amd.pipeline import expand_by_durations
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
    """The engine PL-BERT must not silently run the HF forward on CPU tensors (ST2_BERT=hf is the explicit switch)."""
    import torch
    from _util import manifest
    from styletts2_amd import models
    from styletts2_amd._lib import St2Error
    bert = models.load_plbert(manifest("ljspeech")["plbert"]).eval()
    ids = torch.zeros(1, 5, dtype=torch.long)
    with pytest.raises(St2Error):
        bert(ids, attention_mask=torch.ones(1, 5, dtype=torch.int32))
