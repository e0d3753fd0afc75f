"""tools/validate_checkpoint.py -- the first-contact tool for a real checkpoint (models.py:696-713 loading, one short utterance
on oracle and engine, per-tap errors against the bars) -- driven on the synthetic full-layout checkpoint of
test_checkpoint_layout.py through the CPU backend (C++ plans on host memory).  The GPU leg (two-sided operand table before /
after calibration, status word) is test_calibration_gpu.py's + the tool's own run on a GPU box (profiles/r05*_validate_*.txt)."""
import importlib.util
import io
import os

import pytest

from test_checkpoint_layout import _write_checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("validate_checkpoint", os.path.join(ROOT, "tools", "validate_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("tag", ["ljspeech", "libritts"])
def test_validate_checkpoint_on_the_cpu_backend(tmp_path, tag):
    _, _, path, cfg_path = _write_checkpoint(tmp_path, tag)
    out = io.StringIO()
    res = _tool().validate(path, cfg_path, n_tokens=8, batch=1, steps=3, backend="cpu", out=out)
    text = out.getvalue()
    assert res["ok"], text
    assert res["multispeaker"] == (tag == "libritts") and res["decoder"] == ("hifigan" if tag == "libritts" else "istftnet")
    assert abs(res["sigma_data"] - 0.1734) < 1e-9, "sigma_data comes from the SAVED config"
    assert res["durations_equal"] is True
    assert set(res["taps"]) >= {"t_en", "d", "s_pred", "asr", "F0", "N", "encode", "front"}
    assert res["wave_rms_err"] < 1e-4 and res["mel_l1"] < 1e-3
    assert "every bar met" in text and "device status word: 0x0" in text
