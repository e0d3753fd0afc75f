"""GPU probe: st2_convt_interleave_stats (the finishing pass of every polyphase ConvTranspose1d) at the vocoders' shapes, B = 32:
ms per launch and TB/s of its 12 bytes per output (phases in, added tensor in, output out)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _lib

if len(sys.argv) > 1:  # A-B against another build of the library
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from styletts2_amd import ops  # noqa: E402

print("library:", _lib.LIB_PATH)

dev = "cuda"
B = 32
# (C_out, stride, L_raw, pad, reflect): HiFi-GAN rates 10 / 5 / 3 / 2 (kernel = 2 * stride), iSTFTNet 10 / 6 (+ reflection pad)
cases = [(256, 10, 4000, 5, False), (128, 5, 20000, 2, False), (64, 3, 60000, 1, False), (32, 2, 120000, 1, False),
         (256, 10, 4000, 5, False), (128, 6, 24000, 3, True)]
for C, s, L, pad, refl in cases:
    Lq = L // s + 1
    ph = torch.randn(B, s * C, Lq, device=dev)
    Lo = L + (1 if refl else 0)
    pitch = (Lo + 31) // 32 * 32
    add = torch.randn(B, C, pitch, device=dev)[:, :, :Lo]
    out = torch.empty(B, C, pitch, device=dev)[:, :, :Lo]
    bias = torch.randn(C, device=dev)
    fn = lambda: ops.convt_interleave(ph, C, s, pad, L, bias=bias, add=add, reflect_left=refl, out=out, want_stats=True)
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("C=%d stride=%d L=%d%s: %.3f ms (incl. the statistics finaliser), %.2f TB/s of 12 B / output, checksum %.6e"
          % (C, s, Lo, " reflect" if refl else "", ms, B * C * Lo * 12 / ms / 1e9, float(out.double().sum())), flush=True)
