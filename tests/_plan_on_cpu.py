"""Test helper: run the engine's host-side layer plans on CPU by substituting the per-kernel fp32
contracts (oracle/ops_ref.py) for the HIP wrappers.  This checks plan wiring, weight-norm folding,
polyphase / K-major packing and buffer aliasing without a GPU; the HIP kernels themselves are checked
against the same contracts on the GPU box (tests/test_ops_gpu.py)."""
import contextlib

from oracle import ops_ref
from styletts2_amd import ops

_NAMES = ["conv1d", "conv1d_direct", "phase_split", "instnorm_stats", "colnorm_stats", "style_fc", "convt_interleave",
          "adain_leaky_pool", "har_source", "stft_mag_phase", "istft", "attention", "colnorm_apply", "lstm_bidir", "add_chanvec", "mean_tokens",
          "axpbypcz", "time_features", "tokens_to_channels", "broadcast_cols", "copy_ncl", "duration_head",
          "expand_by_durations", "stft_frames", "power_spectrum", "log_norm_", "dwconv3x3s2", "avgpool2x2"]


@contextlib.contextmanager
def ops_on_cpu():
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, getattr(ops_ref, n))
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
