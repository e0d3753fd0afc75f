"""In-tree build of libst2_hip.so (hipcc, gfx950 only).

`python -m styletts2_amd._build` or `__graft_entry__.build()`.  Objects are cached under
styletts2_amd/csrc/build/ and rebuilt when a source or header is newer.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_PATH = os.path.join(HERE, "libst2_hip.so")

SOURCES = ["st2_api.hip", "st2_conv1d.hip", "st2_conv1d_f16s.hip", "st2_conv1d_f16s_k0.hip", "st2_conv1d_f16s_k1.hip",
           "st2_conv1d_f16s_k2.hip", "st2_conv1d_xs.hip", "st2_conv1d_xs_k0.hip", "st2_conv1d_xs_k1.hip",
           "st2_conv1d_xs_k2.hip", "st2_conv1d_xs_k3.hip", "st2_actsplit.hip", "st2_norm.hip", "st2_misc.hip", "st2_glue.hip", "st2_style.hip", "st2_engine.hip", "st2_source.hip", "st2_attention.hip", "st2_lstm.hip", "st2_lstm_coop.hip"]
# -ffp-contract=off: the SineGen phase path must reproduce ATen-CPU rounding (no implicit FMA);
# fused multiply-adds are written explicitly (fmaf) where wanted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function", "-I" + INCLUDE, "-I" + CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build_lib(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("st2_common.h", "st2_act.h", "st2_conv1d_xs_impl.h",
                                                "st2_conv1d_f16s_impl.h")] + [os.path.join(INCLUDE, "st2.h")]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print("[st2 build]", " ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    if force or procs or _newer(objs, LIB_PATH):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print("[st2 build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
    print(LIB_PATH)
