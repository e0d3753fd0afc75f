"""Fixture generator of the mel front-end: outputs of oracle/mel_ref.py (fp64 evaluation of torchaudio's documented
MelSpectrogram algorithm, see its header) on seeded waveforms -> tests/golden/mel_vectors.npz.  The waveforms are
regenerated from the seeds on the test side (`waves()` below); only the float32 log-mels are stored.

    python -m oracle.golden_mel
"""
import os

import numpy as np
import torch

from oracle import mel_ref

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def waves():
    """name -> float32 waveform tensor [B, L]: noise + a decaying 220 Hz tone (speech-like dynamic range), the same at
    full scale (|X|^2 ~ 1e5), and an odd length that is not a multiple of the hop."""
    g = torch.Generator().manual_seed(5)
    t = torch.arange(48000) / 24000.0
    a = torch.randn(2, 48000, generator=g) * 0.05 + 0.4 * torch.sin(2 * torch.pi * 220.0 * t) * torch.exp(-t)
    b = (a * 2.4).clamp(-1, 1)
    c = torch.randn(1, 7013, generator=g) * 0.1
    return {"tone_noise": a, "full_scale": b, "odd_length": c}


def main():
    out = {k: mel_ref.mel_spectrogram(v.double().numpy()).astype(np.float32) for k, v in waves().items()}
    path = os.path.join(GOLDEN, "mel_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
