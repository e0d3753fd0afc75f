"""GPU probe (round 3, first visit): st2_style_forward (C++ plan, ABI v16) against StyleEncoder.forward (the per-kernel
Python plan) -- same kernels, same arguments: expected bitwise equal.  The plan was validated on the CPU backend only
(tests/test_engine_cpu.py::test_engine_style_plan) because round 2 had no GPU minutes left; once this passes on the GPU it
becomes a test in tests/test_engine_gpu.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import synth
from styletts2_amd import engine, style

bad = 0
for (dim_in, sd, mx, B, T, seed) in ((16, 32, 64, 3, 83, 21), (64, 128, 512, 2, 120, 22), (64, 128, 512, 1, 300, 23)):
    enc = style.StyleEncoder(dim_in=dim_in, style_dim=sd, max_conv_dim=mx).eval()
    synth.init_spectral_norm_(enc, seed)
    enc = enc.to("cuda")
    mel = torch.randn(B, 1, 80, T, generator=torch.Generator().manual_seed(seed)).cuda()
    ref = enc(mel)
    eng = engine.build_style_engine(enc, None, torch.device("cuda"))
    out = eng.style_forward(0, mel)
    torch.cuda.synchronize()
    d = float((out - ref).abs().max())
    print("style plan dim_in=%d T=%d B=%d: max |diff| = %.3e of %.3e (bitwise: %s)" % (dim_in, T, B, d, float(ref.abs().max()),
                                                                                      torch.equal(out, ref)))
    bad |= not torch.equal(out, ref)
sys.exit(1 if bad else 0)
