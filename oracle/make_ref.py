"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- builds oracle/_ref/: the reference's own hot-path modules, compiled to
CPython bytecode from the sources WHERE THEY LIE under /root/reference (no source file is copied into this repository;
oracle/_ref/ is git-ignored build output, like a compiled C reference would be, and travels to the GPU box with gpurun).

    python -m oracle.make_ref            # what __graft_entry__.build() runs when /root/reference exists

Why: /root/reference does not exist on the GPU box, so bench.py's `cpu_baseline` leg could only time the oracle port
there (`"kind": "port"`).  With oracle/_ref/ present it times the UNMODIFIED reference modules (`"kind": "reference"`,
SURVEY.md section 8d: models.py:614-694 + Demo/Inference_LJSpeech.ipynb:268-315) on the GPU box's host cores, the port
beside it.  Only bench.py's cpu_baseline leg and tests/ read oracle/_ref/ (through oracle/ref_harness.py with
ST2_REFERENCE_ROOT pointing here); the product never does.

What is built: the import closure of the harness (every module under /root/reference that `ref_harness.load_reference()`
pulls in: models, Modules/*, Modules/diffusion/*, Utils/ASR, Utils/JDC, Utils/PLBERT/util), each as a sourceless
`<name>.pyc` at the same relative path, and `configs.json` = the parsed Configs/config.yml, Configs/config_libritts.yml
and Utils/PLBERT/config.yml (the values already held by benchdata/manifests).  A `MANIFEST.json` records the source
path, size and sha256 of every input so that a stale _ref is detectable.
"""
import hashlib
import json
import os
import py_compile
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SRC_ROOT = "/root/reference"
CONFIGS = ["Configs/config.yml", "Configs/config_libritts.yml", "Utils/PLBERT/config.yml"]

_CLOSURE = r"""
import json, os, sys
sys.path.insert(0, %r)
os.environ["ST2_REFERENCE_ROOT"] = %r
from oracle import ref_harness as RH
RH.load_reference()
root = os.path.realpath(%r) + os.sep
files = sorted({os.path.realpath(m.__file__) for m in list(sys.modules.values())
                if getattr(m, "__file__", None) and os.path.realpath(m.__file__).startswith(root)})
print("CLOSURE " + json.dumps(files))
"""


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def available():
    return os.path.isfile(os.path.join(OUT, "MANIFEST.json"))


def build(verbose=True):
    if not os.path.isfile(os.path.join(SRC_ROOT, "models.py")):
        if verbose:
            print("[make_ref] %s not present: oracle/_ref left as is (%s)" % (SRC_ROOT, "present" if available() else "absent"))
        return None
    repo = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, "-c", _CLOSURE % (repo, SRC_ROOT, SRC_ROOT)], capture_output=True, text=True,
                         cwd=repo)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("CLOSURE ")]
    if out.returncode != 0 or not line:
        raise RuntimeError("make_ref: importing the reference failed:\n" + out.stderr[-2000:])
    files = json.loads(line[0][len("CLOSURE "):])
    import yaml
    man = {"python": sys.version.split()[0], "source_root": SRC_ROOT, "modules": {}, "configs": {}}
    root = os.path.realpath(SRC_ROOT)
    for src in files:
        rel = os.path.relpath(src, root)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the path recorded in tracebacks / co_filename -- the ORIGINAL location, for file:line citations
        py_compile.compile(src, cfile=dst, dfile=src, doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        man["modules"][rel] = {"bytes": os.path.getsize(src), "sha256": _sha(src)}
    cfgs = {}
    for rel in CONFIGS:
        with open(os.path.join(root, rel)) as f:
            cfgs[rel] = yaml.safe_load(f)
        man["configs"][rel] = {"sha256": _sha(os.path.join(root, rel))}
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump(cfgs, f, sort_keys=True)
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    if verbose:
        print("[make_ref] %d reference modules compiled to bytecode under %s" % (len(files), OUT))
    return OUT


if __name__ == "__main__":
    build()
