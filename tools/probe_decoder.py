"""GPU probe: full-size decoder forward timing (B x 10 s) with a per-kernel breakdown via torch profiler-free
event timing of the whole forward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from _util import decoder_kwargs, manifest
from styletts2_amd import synth
from styletts2_amd.decoder import Decoder

tag = os.environ.get("PROBE_TAG", "ljspeech")
B = int(os.environ.get("PROBE_B", "32"))
T = int(os.environ.get("PROBE_T", "400"))
dc = manifest(tag)["config"]["decoder"]
dec = Decoder(**decoder_kwargs(dc)).eval()
synth.init_synthetic_(dec, 1)
dec = dec.to("cuda")
asr, F0, N, s, noise = [t.to("cuda") for t in synth.decoder_inputs(B, T, 3)]
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    out = dec(asr, F0, N, s, noise=noise)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("iter %d: %.1f ms  -> %.0f audio-s/s  (B=%d, T=%d, %s) finite=%s absmax=%.3f" % (
        it, dt * 1e3, B * T * 600 / 24000 / dt, B, T, tag, bool(torch.isfinite(out).all()), out.abs().max().item()),
        flush=True)
print("max mem GB", torch.cuda.max_memory_allocated() / 1e9)
