"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *unmodified* reference modules from /root/reference so that
(a) golden fixtures can be generated (oracle/make_golden.py) and
(b) the CPU restatement in oracle/st2_oracle.py can be pinned against them.

/root/reference only exists in the build container, never on the GPU box.  What travels there is oracle/_ref/ (built
by oracle/make_ref.py from the reference where it lies: its modules as CPython bytecode, git-ignored): bench.py's
`cpu_baseline` leg loads the reference through this file from there (`"kind": "reference"`).  The `-m gpu` tests and
smoke() do not use it; the product never imports it.

The three sys.modules stubs follow SURVEY.md App. A.5: the reference imports
`einops_exts`, `munch` and `torchaudio`, none of which is installed here and
none of which carries arithmetic on the hot path.
"""
import os
import sys
import types

_BUILT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")  # oracle/make_ref.py: bytecode of the reference


def _default_root():
    """ST2_REFERENCE_ROOT if set; the reference checkout where it exists (build container); else oracle/_ref -- the
    reference's own modules compiled to bytecode by oracle/make_ref.py, which is what travels to the GPU box."""
    env = os.environ.get("ST2_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isfile("/root/reference/models.py"):
        return "/root/reference"
    return _BUILT


REFERENCE_ROOT = _default_root()


def reference_available() -> bool:
    return any(os.path.isfile(os.path.join(REFERENCE_ROOT, n)) for n in ("models.py", "models.pyc"))


def reference_kind() -> str:
    """"source" (a checkout) or "bytecode" (oracle/_ref): the same modules either way."""
    return "source" if os.path.isfile(os.path.join(REFERENCE_ROOT, "models.py")) else "bytecode"


def _install_stubs():
    import transformers  # noqa: F401  must be imported before the torchaudio stub
    from einops import rearrange

    if "einops_exts" not in sys.modules:
        m = types.ModuleType("einops_exts")
        m.rearrange_many = lambda ts, pattern, **kw: tuple(rearrange(t, pattern, **kw) for t in ts)
        sys.modules["einops_exts"] = m

    if "munch" not in sys.modules:
        m = types.ModuleType("munch")

        class Munch(dict):
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError as e:
                    raise AttributeError(k) from e

            def __setattr__(self, k, v):
                self[k] = v

        m.Munch = Munch
        sys.modules["munch"] = m

    if "torchaudio" not in sys.modules:
        import torch

        ta = types.ModuleType("torchaudio")
        taf = types.ModuleType("torchaudio.functional")
        tat = types.ModuleType("torchaudio.transforms")

        def create_dct(n_mfcc, n_mels, norm):
            import math
            n = torch.arange(float(n_mels))
            k = torch.arange(float(n_mfcc)).unsqueeze(1)
            dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
            if norm is None:
                dct *= 2.0
            else:
                dct[0] *= 1.0 / math.sqrt(2.0)
                dct *= math.sqrt(2.0 / float(n_mels))
            return dct.t()

        taf.create_dct = create_dct

        class _Unavailable:
            def __init__(self, *a, **k):
                raise RuntimeError("torchaudio stub: transforms are not on the hot path")

        tat.MelSpectrogram = _Unavailable
        ta.functional = taf
        ta.transforms = tat
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.functional"] = taf
        sys.modules["torchaudio.transforms"] = tat


_loaded = {}


def load_reference():
    """Returns a namespace with the reference's own modules (models, istftnet, hifigan, sampler...)."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import models as ref_models
    import Modules.istftnet as ref_istftnet
    import Modules.hifigan as ref_hifigan
    import Modules.diffusion.sampler as ref_sampler
    import Modules.diffusion.modules as ref_dmodules
    from Utils.PLBERT.util import CustomAlbert
    _loaded.update(models=ref_models, istftnet=ref_istftnet, hifigan=ref_hifigan,
                   sampler=ref_sampler, dmodules=ref_dmodules,
                   CustomAlbert=CustomAlbert)
    return types.SimpleNamespace(**_loaded)


def _yaml(rel):
    path = os.path.join(REFERENCE_ROOT, rel)
    if os.path.isfile(path):
        import yaml
        with open(path) as f:
            return yaml.safe_load(f)
    import json
    with open(os.path.join(REFERENCE_ROOT, "configs.json")) as f:  # oracle/_ref: the parsed ymls (make_ref.py)
        return json.load(f)[rel]


def load_config(name="config.yml"):
    return _yaml("Configs/" + name)


def plbert_config():
    return _yaml("Utils/PLBERT/config.yml")["model_params"]


def recursive_munch(d):
    """Same contract as the reference's utils.recursive_munch (utils.py:63-69); restated here
    because utils.py imports monotonic_align/librosa, which are not installed."""
    Munch = sys.modules["munch"].Munch
    if isinstance(d, dict):
        return Munch((k, recursive_munch(v)) for k, v in d.items())
    if isinstance(d, list):
        return [recursive_munch(v) for v in d]
    return d


def build_reference_model(config_name="config.yml", overrides=None, seed=0, replace_keys=()):
    """build_model(...) from the reference (models.py:614-694) with a random-init PL-BERT.  `overrides` is merged into
    model_params key by key; keys named in `replace_keys` are REPLACED wholesale instead (a decoder block of another
    type must not inherit the old type's list lengths)."""
    import torch
    from transformers import AlbertConfig
    ref = load_reference()
    cfg = load_config(config_name)
    mp = cfg["model_params"]
    if overrides:
        def merge(d, o):
            for k, v in o.items():
                if isinstance(v, dict) and isinstance(d.get(k), dict):
                    merge(d[k], v)
                else:
                    d[k] = v
        for k in replace_keys:
            mp[k] = overrides[k]
        merge(mp, {k: v for k, v in overrides.items() if k not in replace_keys})
    args = recursive_munch(mp)
    torch.manual_seed(seed)
    bert = ref.CustomAlbert(AlbertConfig(**plbert_config()))
    model = ref.models.build_model(args, None, None, bert)
    for k in model:
        if model[k] is not None:
            model[k].eval()
    return model, args, cfg
