#!/usr/bin/env python
"""Is the hipMemsetAsync of st2_lstm_bidir_coop replayed by a captured graph?  The cooperative launch under torch.cuda.graph on a
scratch buffer pre-filled with 0x5A: after replay scratch[0] must be 0 (and the granule tags those of THIS run)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from styletts2_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
H = 256
torch.manual_seed(0)
whh = (torch.randn(2, H, 4 * H, device=dev) / 16).contiguous()
for B, N in ((1, 96), (32, 100)):
    G = torch.randn(B, 8 * H, N, device=dev)
    nbytes = lib.st2_lstm_coop_scratch_bytes(B)
    Y = torch.empty(B, 2 * H, N, device=dev)
    scratch = torch.full((nbytes,), 0x5A, device=dev, dtype=torch.uint8)

    def call(fn):
        rc = fn(G.data_ptr(), G.stride(0), G.stride(1), whh.data_ptr(), 0, B, H, N, Y.data_ptr(), Y.stride(0), Y.stride(1),
                scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.st2_last_error()
    for name, fn in (("coop", lib.st2_lstm_bidir_coop), ("recovering", lib.st2_lstm_bidir_coop_recovering)):
        call(fn)  # eager once (attributes, status word)
        torch.cuda.synchronize()
        ref = Y.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            call(fn)
        for it in range(3):
            scratch.fill_(0x5A)
            Y.fill_(7.0)
            ops.status(clear=True)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            print("B %2d %-10s replay %d: scratch[0] = 0x%x, scratch[1] = 0x%x, status 0x%x, Y equal eager: %s" % (
                B, name, it, int(scratch[:4].view(torch.int32).item()) & 0xffffffff, int(scratch[4:8].view(torch.int32).item()) & 0xffffffff,
                ops.status(clear=True), torch.equal(Y, ref)), flush=True)
