"""Workload definitions shared by bench.py, __graft_entry__.smoke(), the parity tests and the oracle's fixture
generators: the model manifests (config + state_dict layout, extracted from the reference by oracle/make_golden.py) and
the seeded synthetic weights / inputs (`synth`).  No product code and no test code lives here; nothing here touches the
HIP library."""
import json
import os

MANIFESTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manifests")


def manifest(tag):
    """{"config": ..., "plbert": ..., "state_dicts": ...} of one reference configuration (ljspeech / libritts /
    libritts_istftnet)."""
    with open(os.path.join(MANIFESTS, "manifest_%s.json" % tag)) as f:
        return json.load(f)
