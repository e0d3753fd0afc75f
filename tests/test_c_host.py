"""A non-Python host on the module-level C ABI: tools/st2_c_host.c -- plain C, include/st2.h + the
HIP runtime only -- is compiled with gcc, fed a weight / input bundle, and its waveform is compared with the Python
binding's on the same weights and inputs (both drive the same C++ launch plan: bitwise equal)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from _util import decoder_kwargs, manifest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "st2_c_host.c")
LIBDIR = os.path.join(ROOT, "styletts2_amd")


def _build(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc or the HIP headers are not installed")
    if not os.path.exists(os.path.join(LIBDIR, "libst2_hip.so")):
        pytest.skip("libst2_hip.so not built (run __graft_entry__.build())")
    exe = os.path.join(str(tmp_path), "st2_c_host")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + LIBDIR, "-lst2_hip", "-L/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_c_host_builds_against_the_header(tmp_path):
    """CPU box: the C translation unit compiles as C11 against include/st2.h and links against libst2_hip.so (every
    module-level symbol it uses is exported)."""
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("tag,B,T", [("ljspeech", 2, 24), ("libritts", 1, 31)])
def test_c_host_matches_python_binding_bitwise(tmp_path, tag, B, T):
    from styletts2_amd import engine
    import synth  # tests/synth.py: seeded synthetic weights / inputs (test + bench helper, not product code)
    from styletts2_amd.decoder import Decoder
    exe = _build(tmp_path)
    dc = manifest(tag)["config"]["decoder"]
    dec = Decoder(**decoder_kwargs(dc)).eval()
    synth.init_synthetic_(dec, 4)
    asr, F0, N, s, noise = synth.decoder_inputs(B, T, 9)
    cfg = engine.decoder_config(dec)
    bundle = os.path.join(str(tmp_path), "bundle.bin")
    with open(bundle, "wb") as f:
        f.write(bytes(cfg))
        f.write(struct.pack("<ii", B, T))
        state = engine._folded_state(dec)
        f.write(struct.pack("<i", len(state)))
        for name, t in state.items():
            nm = ("decoder." + name).encode()
            f.write(struct.pack("<i", len(nm)) + nm)
            f.write(struct.pack("<i", t.dim()) + struct.pack("<%dq" % t.dim(), *t.shape))
            f.write(t.contiguous().numpy().astype("<f4").tobytes())
        for t in (asr, F0, N, s, noise):
            f.write(t.contiguous().numpy().astype("<f4").tobytes())
    out = os.path.join(str(tmp_path), "wave.bin")
    r = subprocess.run([exe, bundle, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    wave_c = torch.from_numpy(np.fromfile(out, dtype="<f4").copy()).reshape(B, 1, 600 * T)
    dec = dec.to("cuda")
    os.environ.pop("ST2_PLAN", None)
    wave_py = dec(asr.cuda(), F0.cuda(), N.cuda(), s.cuda(), noise=noise.cuda()).cpu()
    assert bool(torch.isfinite(wave_c).all())
    assert torch.equal(wave_c, wave_py)
