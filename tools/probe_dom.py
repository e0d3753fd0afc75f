"""GPU probe for the PMC passes: the bench's dominant launch class (C=128, L=48001, k=11 resblock convs at B=32,
AdaIN+Snake prologue) in the mix one AdaINResBlock1 issues -- convs1 with dilation 1/3/5 (no residual) and three
convs2 (dilation 1, residual epilogue) -- next to two calibration kernels with a known byte count (a 786 MB device
copy = read + write; st2_instnorm_stats = read only).  Also times the class with HIP events (printed)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import ops, weights

dev = "cuda"
B, Cc, L, ks = int(os.environ.get("PROBE_B", "32")), 128, 48001, 11
reps = int(os.environ.get("PROBE_REPS", "2"))
g = torch.Generator(device=dev).manual_seed(0)
# PROBE_PITCH=1 (default): rows start 128-byte aligned (pitch = L rounded up to 32 floats), the layout of the C++ plan's
# workspace (st2_engine.hip pitch_of); 0 = dense rows (L = 48001 is odd: every row start is misaligned)
pitch = (L + 31) // 32 * 32 if os.environ.get("PROBE_PITCH", "1") == "1" else L
x = torch.randn(B, Cc, pitch, device=dev, generator=g)[:, :, :L]
w = torch.randn(Cc, Cc, ks, device=dev, generator=g) / math.sqrt(Cc * ks)
wt = weights.pack_conv_f16s(w).to(dev)
bias = torch.randn(Cc, device=dev, generator=g)
h = torch.randn(B, 2 * Cc, device=dev, generator=g) * 0.3
alpha = torch.rand(Cc, device=dev, generator=g) + 0.5
out = torch.empty((B, Cc, pitch), device=dev)[:, :, :L]
st = ops.instnorm_stats(x)


def resblock_mix():
    for dil in (1, 3, 5):
        ops.conv1d(x, wt, Cc, ks, dil=dil, pad_left=(ks - 1) * dil // 2, bias=bias, out=out, pro=ops.PRO_ADAIN_SNAKE,
                   stats=st, gamma=h[:, :Cc], beta=h[:, Cc:], alpha=alpha)
        ops.conv1d(x, wt, Cc, ks, dil=1, pad_left=(ks - 1) // 2, bias=bias, out=out, pro=ops.PRO_ADAIN_SNAKE,
                   stats=st, gamma=h[:, :Cc], beta=h[:, Cc:], alpha=alpha, res=x)


resblock_mix()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    resblock_mix()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (6 * reps)
print("probe_dom: row pitch %d floats" % pitch)
print("probe_dom: B=%d mean launch %.4f ms, %.1f algorithmic TFLOP/s" % (B, ms, 2.0 * B * Cc * Cc * ks * L / ms / 1e9))
xc, dst = x.contiguous(), torch.empty((B, Cc, L), device=dev)  # calibration kernels on dense tensors of known size
torch.cuda.synchronize()
for _ in range(reps):
    dst.copy_(xc)
    ops.instnorm_stats(xc, out=st)
torch.cuda.synchronize()
print("probe_dom: bytes(x)=%d" % (x.numel() * 4))
