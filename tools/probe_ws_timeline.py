"""GPU probe: phase timeline of workgroup 0 of the warp-specialised fused conv (measurement build with -DST2_WS_TIMELINE:
    WS_EXTRA=-DST2_WS_TIMELINE WS_SUFFIX=_tl bash tools/build_ws_ablate.sh 0).  Four s_memtime stamps per phase and role:
consumer: phase start, (same), end of the k loop / epilogue, after the barrier; producer: phase start, loads issued, chunk staged,
after the barrier."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from styletts2_amd import _hooks, _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "tools/bin/libst2_hip_ws_abl0_tl.so")
from styletts2_amd import ops, weights  # noqa: E402

dev = "cuda"
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.st2_debug_ws_timeline.restype = C.c_int
raw.st2_debug_ws_timeline.argtypes = [C.c_void_p]
_hooks.conv_path = "fused"
lib.st2_conv1d_f16s_set_variant(2)
for (B, Ci, Co, L, ks, dil) in [(32, 64, 64, 120000, 11, 5), (32, 64, 64, 120000, 3, 1), (32, 128, 128, 48001, 3, 1)]:
    pitch = (L + 31) // 32 * 32
    x = torch.randn(B, Ci, pitch, device=dev)[:, :, :L]
    w = torch.randn(Co, Ci, ks, device=dev) / math.sqrt(Ci * ks)
    wt = weights.pack_conv_f16s(w).to(dev)
    bias = torch.randn(Co, device=dev)
    st = ops.instnorm_stats(x)
    h = torch.randn(B, 2 * Ci, device=dev) * 0.3
    alpha = torch.rand(Ci, device=dev) + 0.5
    pad = (ks - 1) * dil // 2
    out = torch.empty((B, Co, pitch), device=dev)[:, :, :L]
    for _ in range(3):
        ops.conv1d(x, wt, Co, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x, pro=ops.PRO_ADAIN_SNAKE, stats=st,
                   gamma=h[:, :Ci], beta=h[:, Ci:], alpha=alpha)
    torch.cuda.synchronize()
    buf = np.zeros(12 * 1024, dtype=np.uint64)
    n = raw.st2_debug_ws_timeline(buf.ctypes.data)
    ops.conv1d(x, wt, Co, ks, dil=dil, pad_left=pad, bias=bias, out=out, res=x, pro=ops.PRO_ADAIN_SNAKE, stats=st,
               gamma=h[:, :Ci], beta=h[:, Ci:], alpha=alpha)
    n = raw.st2_debug_ws_timeline(buf.ctypes.data)
    tl = buf.reshape(12, n).astype(np.int64)
    NW = int((tl[:, 0] != 0).sum())
    tl = tl[:NW]
    hw = tl[:, 0]
    print("== C=%d->%d L=%d k=%d dil=%d" % (Ci, Co, L, ks, dil))
    print("wave -> HW_ID: " + ", ".join("w%d: simd %d slot %d cu %d (0x%x)" % (w, (hw[w] >> 4) & 3, hw[w] & 15, (hw[w] >> 8) & 15, hw[w])
                                         for w in range(NW)))
    st_ = tl[:, 1:]
    nph = int(min((st_[w] > 0).sum() for w in range(NW))) // 4
    t00 = st_[:, 0].min()
    print("phase | per wave (consumers w0-3, producers w4-7): work / drain of its memory operations / barrier | phase length")
    for i in range(20, min(nph, 32)):
        cols = ["%5d/%5d/%5d" % (st_[w, 4 * i + 1] - st_[w, 4 * i], st_[w, 4 * i + 2] - st_[w, 4 * i + 1],
                                 st_[w, 4 * i + 3] - st_[w, 4 * i + 2]) for w in range(NW)]
        end = max(st_[w, 4 * i + 3] for w in range(NW))
        start = min(st_[w, 4 * i] for w in range(NW))
        print("%5d | %s | %6d" % (i, " ".join(cols), end - start))
    tot = [sum(st_[w, 4 * i + 1] - st_[w, 4 * i] for i in range(nph)) for w in range(NW)]
    print("work totals over %d phases: %s; span %d ticks" % (nph, tot, st_[0, 4 * nph - 1] - t00))
