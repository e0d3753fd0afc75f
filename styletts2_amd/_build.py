"""In-tree build of libst2_hip.so (hipcc, gfx950 only).

`python -m styletts2_amd._build` or `__graft_entry__.build()`.  Objects are cached under
styletts2_amd/csrc/build/ next to the compiler-written dependency file of each translation unit (`-MD -MF`): an object is
rebuilt when its source, ANY header it actually included, or the flag set changed -- no hand-kept header list to go stale.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(HERE, "libst2_hip.so")

SOURCES = ["st2_api.hip", "st2_conv1d.hip", "st2_conv1d_f16s.hip", "st2_conv1d_f16s_k0.hip", "st2_conv1d_f16s_k1.hip",
           "st2_conv1d_f16s_k2.hip", "st2_conv1d_f16s_w0.hip", "st2_conv1d_f16s_w1.hip", "st2_conv1d_f16s_w2.hip",
           "st2_conv1d_xs.hip", "st2_conv1d_xs_k0.hip", "st2_conv1d_xs_k1.hip",
           "st2_conv1d_xs_k2.hip", "st2_conv1d_xs_k3.hip", "st2_actsplit.hip", "st2_norm.hip", "st2_misc.hip", "st2_glue.hip", "st2_style.hip", "st2_engine.hip", "st2_source.hip", "st2_attention.hip", "st2_lstm.hip", "st2_lstm_coop.hip", "st2_probe.hip", "st2_probe_conv.hip"]
# -ffp-contract=off: the SineGen phase path must reproduce ATen-CPU rounding (no implicit FMA);
# fused multiply-adds are written explicitly (fmaf) where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function", "-I" + INCLUDE, "-I" + CSRC]
# The fused conv's prologue is VALU-bound and shares its SIMD with the MFMA stream: SLP-packed f32 (v_pk_mul / v_pk_fma with
# the v_mov shuffles that feed them) is an anti-lever there (MI355X_MICROARCH.md; profiles/archive/r03/r03D_probe_narrow_libs.log: one-role
# kernel -4 % at k = 7 / C = 64, -13 % at k = 3 / C = 128 / L = 40 000, the warp-specialised k = 3 / C = 32 build +16 %: not for that one).
EXTRA_FLAGS = {s: ["-fno-slp-vectorize"] for s in SOURCES if s.startswith("st2_conv1d_f16s_k")}
# The BiLSTM recurrences: SLP packing of {a*h, b*h} with h an ODD element of an LDS vector load becomes `v_pk_fma_f32 ...
# op_sel:[0,1,0]`, which on gfx950 returns a wrong low half in lanes 48-63 next to MFMA waves of another queue (round 6:
# tools/simd_hazard_repro.hip).  No auto-packing there; the cooperative kernel packs along K by hand
# (plain encoding).  tools/check_isa.py (run below after every link) keeps the encoding out of the WHOLE library.
EXTRA_FLAGS.update({"st2_lstm.hip": ["-fno-slp-vectorize"], "st2_lstm_coop.hip": ["-fno-slp-vectorize"]})


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(s)) or os.path.getmtime(s) > t for s in src_list)


def _depfile_inputs(depfile):
    """Prerequisites listed in a make-style depfile written by `hipcc -MD -MF` (None if it is missing or unreadable:
    the object is then rebuilt)."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    if ":" not in text:
        return None
    deps = text.split(":", 1)[1].split()
    return [d for d in deps if not d.startswith("/opt/rocm") and not d.startswith("/usr/")] or None


def _stale(src, obj, flags):
    """True when `obj` must be rebuilt: missing, built with other flags, or older than the source / any included header."""
    stamp = obj + ".flags"
    try:
        if open(stamp).read() != " ".join(flags):
            return True
    except OSError:
        return True
    deps = _depfile_inputs(obj[:-2] + ".d")
    return deps is None or _newer([src] + deps, obj)


def build_lib(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        flags = FLAGS + EXTRA_FLAGS.get(s, [])
        if force or _stale(src, obj, flags):
            cmd = [hipcc] + flags + ["-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
            if verbose:
                print("[st2 build]", " ".join(cmd), flush=True)
            if os.path.exists(obj + ".flags"):
                os.remove(obj + ".flags")
            procs.append((s, obj, flags, subprocess.Popen(cmd)))
    for s, obj, flags, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
        with open(obj + ".flags", "w") as f:
            f.write(" ".join(flags))
    if force or procs or _newer(objs, LIB_PATH):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print("[st2 build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        isa_gate(verbose)
    return LIB_PATH


def isa_gate(verbose=True):
    """tools/check_isa.py on the linked library: raises when a packed-f32 op with op_sel is present (see EXTRA_FLAGS above)."""
    import importlib.util
    path = os.path.join(os.path.dirname(HERE), "tools", "check_isa.py")
    spec = importlib.util.spec_from_file_location("st2_check_isa", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ok, report = mod.check(LIB_PATH)
    if verbose or ok is False:
        print("[st2 build] " + report, flush=True)
    if ok is False:
        os.replace(LIB_PATH, LIB_PATH + ".rejected")  # never leave a library that failed the gate where the loader finds it
        raise RuntimeError("libst2_hip.so failed the ISA gate (tools/check_isa.py); kept as %s.rejected" % LIB_PATH)
    return ok


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB_PATH)
