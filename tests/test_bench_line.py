"""The bench line stays parseable: < 8 KB, strict JSON, the contract's fields present (the 25 KB line of
round 5 came back `parsed: null` from the driver).  Inputs: the fat lines round 5 committed under profiles/ and a synthetic
worst case; benchdata/line.py is pure Python."""
import glob
import json
import os

import pytest

from benchdata import line as benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _strict(text):
    def bad(tok):
        raise ValueError("non-standard JSON constant %r" % tok)
    return json.loads(text, parse_constant=bad)


def _fat_lines():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r05", "r05[u-y]_bench*.json"))):
        txt = open(f).read().strip().splitlines()[-1]
        if len(txt) > 8192:
            out.append((os.path.basename(f), json.loads(txt)))
    return out


FAT = _fat_lines()


def test_round5_lines_exist_and_were_too_big():
    assert len(FAT) >= 3 and all(len(json.dumps(r)) > 20000 for _, r in FAT)


@pytest.mark.parametrize("name,res", FAT, ids=[n for n, _ in FAT])
def test_fat_lines_compact_below_the_limit(name, res):
    line = benchline.dumps(res, "bench_detail.json")
    assert len(line.encode()) < 8192 and "\n" not in line
    back = _strict(line)
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == pytest.approx(res["value"], rel=1e-4) and back["ms_per_step"] == pytest.approx(res["ms_per_step"], rel=1e-4)
    assert back["config"]["workload"] and "model" not in back["config"]
    roof = back["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
    assert roof["unit"] in ("GB/s", "TFLOP/s") and "traffic" in roof
    if "other_configs" in res:
        for k, v in res["other_configs"].items():
            got = back["other_configs"][k]
            for f in ("ms_per_step", "audio_s_per_s", "latency_ms"):
                if f in v:
                    assert got[f] == v[f]
    assert back["detail"] == "bench_detail.json"


def test_worst_case_still_fits_and_is_strict_json():
    name, res = FAT[0]
    res = json.loads(json.dumps(res))
    res["roofline"]["classes"] = res["roofline"]["classes"] * 40
    res["roofline"]["frac"] = float("nan")  # a NaN must not reach the line as a bare token
    res["config"]["workload"] = "w" * 5000
    res["cpu_baseline"] = {"value": 23.6, "unit": "audio-s/s", "cores": 16, "kind": "reference", "sample": "s" * 3000,
                           "threads_tried": {str(i): 1.0 for i in range(200)}, "host_cores": 256}
    res["other_configs"] = {("leg%d" % i): dict(res["other_configs"]["libritts_hifigan"]) for i in range(40)}
    line = benchline.dumps(res, "bench_detail.json")
    assert len(line.encode()) < 8192
    back = _strict(line)
    assert back["roofline"]["frac"] is None and back["cpu_baseline"]["kind"] == "reference"
    for k in CONTRACT:
        assert k in back


def test_bench_py_prints_through_the_compactor():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "print(json.dumps(res)" not in src and "benchline.dumps(res" in src
