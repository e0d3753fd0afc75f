#!/usr/bin/env python
"""Does st2_lstm_bidir_coop_recovering re-run calls that did not time out?  Per shape: scratch[0] after the call, the sticky
status word, and the time of the pair against the bare cooperative launch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from styletts2_amd import _lib, ops  # noqa: E402

lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)
H = 256
whh = (torch.randn(2, H, 4 * H, device=dev) / 16).contiguous()
for B, N, ragged in ((1, 100, False), (1, 400, False), (1, 96, True), (8, 50, False), (32, 100, False), (32, 400, False), (32, 100, True)):
    G = torch.randn(B, 8 * H, N, device=dev)
    lens = None
    if ragged:
        lens = torch.randint(max(1, N - 15), N + 1, (B,), dtype=torch.int32, device=dev)
    lp = 0 if lens is None else lens.data_ptr()
    nbytes = lib.st2_lstm_coop_scratch_bytes(B)
    out = {}
    for name, fn in (("coop", lib.st2_lstm_bidir_coop), ("recovering", lib.st2_lstm_bidir_coop_recovering)):
        Y = torch.empty(B, 2 * H, N, device=dev)
        scratch = torch.full((nbytes,), 0x5A, device=dev, dtype=torch.uint8)
        ops.status(clear=True)
        st0 = []
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            rc = fn(G.data_ptr(), G.stride(0), G.stride(1), whh.data_ptr(), lp, B, H, N, Y.data_ptr(), Y.stride(0), Y.stride(1),
                    scratch.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, lib.st2_last_error()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20 * 1e6
        out[name] = (int(scratch[:4].view(torch.int32).item()), ops.status(clear=True), dt, Y)
    same = torch.equal(out["coop"][3], out["recovering"][3])
    print("B %2d N %3d ragged %d: coop scratch[0] %d status 0x%x %.0f us | recovering scratch[0] %d status 0x%x %.0f us | outputs equal %s" % (
        B, N, ragged, out["coop"][0], out["coop"][1], out["coop"][2], out["recovering"][0], out["recovering"][1], out["recovering"][2], same))
