"""GPU probe: st2_istft against the contract (torch.istft), error per 256-sample tile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import ops_ref as R
from styletts2_amd import ops

g = torch.Generator().manual_seed(0)
for (B, M) in ((2, 201), (1, 2001), (3, 57)):
    sp = torch.cat([torch.rand(B, 11, M, generator=g) * 2, torch.rand(B, 11, M, generator=g) * 6 - 3], dim=1)
    ref = R.istft(sp, 20, 5)
    out = ops.istft(sp.cuda(), 20, 5).cpu()
    d = (out - ref).abs()
    print("B=%d M=%d max err %.3e" % (B, M, d.max().item()))
    if d.max() > 1e-4:
        per = d[0, 0]
        for t0 in range(0, per.numel(), 256):
            blk = per[t0:t0 + 256]
            bad = (blk > 1e-4).nonzero().flatten()
            print("  tile %d: max %.3e, bad %d first %s" % (t0 // 256, blk.max().item(), bad.numel(), bad[:8].tolist()))
            if t0 > 1024:
                break
