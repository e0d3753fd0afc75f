"""GPU probe: the two BiLSTM recurrence kernels at the bench shapes (B=32; N=100 text side, T=400 prosody side)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _hooks, _lib, ops

dev = "cuda"
B, H = int(os.environ.get("PROBE_B", "32")), 256
whh = (torch.randn(2, H, 4 * H, device=dev) / 16).contiguous()
for N in (100, 400):
    G = torch.randn(B, 8 * H, N, device=dev)
    outs = {}
    for mode in ("single", "coop_fence", "coop_sc1", "coop"):  # coop = tagged 8-byte granules (the default hand-off)
        _hooks.lstm = "single" if mode == "single" else "coop"
        _lib.load().st2_lstm_coop_set_exchange({"coop_fence": 0, "coop_sc1": 1}.get(mode, 2))
        for _ in range(2):
            y = ops.lstm_bidir(G, whh)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = ops.lstm_bidir(G, whh)
        e1.record()
        torch.cuda.synchronize()
        outs[mode] = y
        print("lstm %s B=%d N=%d: %.3f ms (%.2f us/step) status=%d" % (mode, B, N, e0.elapsed_time(e1) / 5,
                                                                      e0.elapsed_time(e1) / 5 / N * 1e3,
                                                                      ops.lstm_coop_status() if mode != "single" else 0),
              flush=True)
    _lib.load().st2_lstm_coop_set_exchange(2)
    for blk in (2, 4, 8):  # utterances per cooperative group: 4 -> twice the groups (128 workgroups at B = 32), half the mat-vec
        _lib.load().st2_lstm_coop_set_block(blk)
        _hooks.lstm = "coop"
        for _ in range(2):
            y = ops.lstm_bidir(G, whh)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = ops.lstm_bidir(G, whh)
        e1.record()
        torch.cuda.synchronize()
        print("lstm coop block=%d B=%d N=%d: %.3f ms (%.2f us/step) max |y - single| = %.3g" % (
            blk, B, N, e0.elapsed_time(e1) / 5, e0.elapsed_time(e1) / 5 / N * 1e3, (y - outs["single"]).abs().max().item()),
            flush=True)
    _lib.load().st2_lstm_coop_set_block(0)
    print("   max |coop - single| = %.3g, |coop_fence - single| = %.3g" % (
        (outs["coop"] - outs["single"]).abs().max().item(), (outs["coop_fence"] - outs["single"]).abs().max().item()),
        flush=True)
