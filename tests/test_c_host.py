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
    from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
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
    wave_py = dec(asr.cuda(), F0.cuda(), N.cuda(), s.cuda(), noise=noise.cuda()).cpu()
    assert bool(torch.isfinite(wave_c).all())
    assert torch.equal(wave_c, wave_py)


TTS_SRC = os.path.join(ROOT, "tools", "st2_c_tts.c")


def _build_tts(tmp_path):
    global SRC
    keep, SRC = SRC, TTS_SRC
    try:
        exe = _build(tmp_path)
    finally:
        SRC = keep
    return exe


def test_c_tts_host_builds_against_the_header(tmp_path):
    """CPU box: the text -> waveform C host compiles as C11 against include/st2.h and links (st2_front_forward,
    st2_prosody_forward, st2_decoder_forward, st2_sampler_table are exported)."""
    exe = _build_tts(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("tag,N", [("ljspeech", 9), ("libritts", 11)])
def test_c_tts_host_matches_python_bitwise(tmp_path, tag, N):
    """Phoneme ids -> waveform by tools/st2_c_tts.c (plain C: st2_front_forward, one read-back of the durations,
    st2_prosody_forward, st2_decoder_forward) == the same three calls through the Python binding == pipeline.inference
    (whose stages are those three calls), on the same weights, tokens and noise."""
    from styletts2_amd import engine, models, pipeline
    from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
    exe = _build_tts(tmp_path)
    man = manifest(tag)
    model = models.build_model(models.recursive_munch(man["config"]), None, None, models.load_plbert(man["plbert"]))
    for i, k in enumerate(["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]):
        synth.init_synthetic_(model[k], 10 + i)
        model[k].eval()
    multi, hifigan = bool(man["config"]["multispeaker"]), man["config"]["decoder"]["type"] == "hifigan"
    B, steps, tail = 1, 3, 0 if multi else 5
    T_max = 50 * N + tail
    g = torch.Generator().manual_seed(21)
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    ref_s = torch.randn(B, 256, generator=g) if multi else None
    pool = torch.randn(B, 600 * T_max, 9, generator=g)
    sampler = models.make_sampler(model)
    table, sigma0 = sampler.step_table(steps)
    cfg = engine.model_config(model)
    bundle = os.path.join(str(tmp_path), "tts.bin")
    f32 = lambda t: t.contiguous().numpy().astype("<f4").tobytes()
    with open(bundle, "wb") as f:
        f.write(bytes(cfg))
        f.write(struct.pack("<8i", B, N, steps, tail, int(hifigan), int(multi), T_max, 1))
        f.write(struct.pack("<3d", 1.0, 0.3, 0.7))
        f.write(struct.pack("<d", sigma0) + struct.pack("<%dd" % len(table), *table))
        state = engine.model_state(model)
        f.write(struct.pack("<i", len(state)))
        for name, t in state.items():
            nm = name.encode()
            f.write(struct.pack("<i", len(nm)) + nm)
            f.write(struct.pack("<i", t.dim()) + struct.pack("<%dq" % t.dim(), *t.shape))
            f.write(f32(t))
        f.write(tokens.numpy().astype("<i8").tobytes())
        f.write(f32(noise) + f32(step_noise))
        if multi:
            f.write(f32(ref_s))
        f.write(f32(pool))
    out = os.path.join(str(tmp_path), "tts_out.bin")
    r = subprocess.run([exe, bundle, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = np.fromfile(out, dtype=np.uint8)
    dur_c = torch.from_numpy(raw[:B * N * 8].view("<i8").copy()).reshape(B, N)
    T = int(dur_c.sum())
    wave_c = torch.from_numpy(raw[B * N * 8:].view("<f4").copy()).reshape(B, 1, 600 * T)
    assert 0 < T <= T_max and bool(torch.isfinite(wave_c).all())
    # the same three calls through the Python binding
    for k in ("decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"):
        model[k].to("cuda")
    d = lambda t: None if t is None else t.cuda()
    eng = engine.build_model_engine(model, torch.device("cuda"))
    fr = eng.front_forward(d(tokens), d(noise), d(step_noise), table, sigma0, ref_s=d(ref_s), tail=tail)
    assert torch.equal(fr["durations"].cpu(), dur_c)
    asr, F0, Nn = eng.prosody_forward(fr["d_cm"], fr["t_en"], fr["durations"], fr["s"], T, shift=hifigan)
    sine = pool[:, :600 * T].cuda()
    wave_py = eng.decoder_forward(asr, F0, Nn, fr["ref"], noise=sine).cpu()
    assert torch.equal(wave_c, wave_py)
    # ... and the product pipeline with the C++ front
    wave_pl = pipeline.inference(model, sampler, d(tokens), None, d(noise), diffusion_steps=steps, ref_s=d(ref_s),
                                 step_noise=d(step_noise), sine_noise=sine, lj_tail=not multi)
    wave_pl = wave_pl[0] if isinstance(wave_pl, list) else wave_pl
    diff = float((wave_pl.cpu().reshape(-1) - wave_c.reshape(-1)).abs().max())
    print("pipeline.inference vs C host: max |diff| = %.3e" % diff)
    assert diff == 0.0
