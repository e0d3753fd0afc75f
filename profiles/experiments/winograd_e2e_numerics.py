"""CPU numerics study, part 3 (no GPU): what does the Toom-Cook F(3,3) / F(4,4) / F(6,6) forms of the k >= 7, C >= 128 convs do to the
END-TO-END waveform of the iSTFTNet generator?  The oracle's generator (oracle/st2_oracle.py) is run three times on the same
synthetic weights and inputs with `F.conv1d` intercepted for the qualifying resblock convs:
  exact     the oracle as it is (fp32 ATen conv)
  direct    this library's operand scheme emulated: 8 x activation and row-scaled weight split into hi + lo f16, three
            products, fp32 accumulation
  F(M,R)    the same operand scheme in the transform domain (input transform in fp32 before the split, weights transformed in
            fp64 at pack time, inverse transform in fp32), dilated layers as stride-d subsequences
and the waveforms are compared at the 1e-4 RMS bar of BASELINE.json's north_star.  Output: profiles/archive/r02/r02_winograd_e2e.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as TF

from _util import decoder_kwargs, manifest, rms
from oracle import st2_oracle as O
from styletts2_amd.decoder import Decoder
from benchdata import synth


def split(a):
    hi = a.half().float()
    return hi, (a - hi).half().float()


def mm3(w, x):  # [co, ci] x [ci, n] with hi / lo operands, fp32 accumulation
    wh, wl = split(w)
    xh, xl = split(x)
    return wh @ xh + wh @ xl + wl @ xh


def row_scale(w2d):
    amax = w2d.abs().amax(dim=1).clamp_min(1e-30)
    _, e = torch.frexp(amax)
    return torch.ldexp(torch.ones_like(amax), 14 - e)


def toom(M, R):
    pts = {3: [0, 1, -1, 2], 4: [0, 1, -1, 2, -2, 0.5], 6: [0, 1, -1, 2, -2, 0.5, -0.5, 3, -3, 1.0 / 3]}[M]
    n = M + R - 1
    AT, G, BT = np.zeros((M, n)), np.zeros((n, R)), np.zeros((n, n))
    for k, p in enumerate(pts):
        AT[:, k] = [p ** i for i in range(M)]
        G[k, :] = [p ** i for i in range(R)]
        G[k, :] /= np.prod([p - q for j, q in enumerate(pts) if j != k])
    AT[M - 1, n - 1] = 1.0
    G[n - 1, R - 1] = 1.0
    Mx = np.array([1.0])
    for q in pts:
        Mx = np.convolve(Mx, np.array([-q, 1.0]))
    for k in range(len(pts)):
        poly = np.array([1.0])
        for j, q in enumerate(pts):
            if j != k:
                poly = np.convolve(poly, np.array([-q, 1.0]))
        BT[k, :len(poly)] = poly
    BT[n - 1, :len(Mx)] = Mx
    return AT, G, BT


def conv_direct(x, w, bias, dil, pad):
    B, C, L = x.shape
    Co, _, K = w.shape
    sc = row_scale(w.reshape(Co, -1))
    xp = TF.pad(x * 8.0, (pad, pad))
    y = torch.zeros(B, Co, L)
    for b in range(B):
        for j in range(K):
            y[b] += mm3(w[:, :, j] * sc[:, None], xp[b, :, j * dil:j * dil + L])
    return y / 8.0 / sc[None, :, None] + bias[None, :, None]


def conv_toom(x, w, bias, dil, pad, M, R):
    AT, G, BT = toom(M, R)
    n = M + R - 1
    B, C, L = x.shape
    Co, _, K = w.shape
    ng = (K + R - 1) // R
    wp = torch.zeros(Co, C, ng * R, dtype=torch.float64)
    wp[:, :, :K] = w.double()
    U = torch.einsum("kr,ocgr->gkoc", torch.from_numpy(G), wp.reshape(Co, C, ng, R)).float()       # [ng][n][Co][C]
    sc = row_scale(U.permute(2, 0, 1, 3).reshape(Co, -1))
    y = torch.zeros(B, Co, L)
    padq = (K - 1) // 2
    BTt, ATt = torch.from_numpy(BT).float(), torch.from_numpy(AT).float()
    for b in range(B):
        for r in range(dil):  # stride-d subsequence a_r[q] = a[d q + r]
            a = x[b, :, r::dil] * 8.0
            Lq = a.shape[1]
            nt = (Lq + M - 1) // M
            ap = TF.pad(a, (padq, M * (nt + ng) + n - Lq))
            idx = (torch.arange(nt + ng) * M)[None, :] + torch.arange(n)[:, None]                   # window of tile T': M T' + n
            V = torch.einsum("kn,cnt->kct", BTt, ap[:, idx])                                        # [n][C][nt + ng]
            Y = torch.zeros(n, Co, nt)
            for g in range(ng):
                for k in range(n):
                    Y[k] += mm3(U[g, k] * sc[:, None], V[k][:, g:g + nt])
            o = torch.einsum("mk,kot->otm", ATt, Y).reshape(Co, nt * M)[:, :Lq]
            y[b, :, r::dil] = o / 8.0 / sc[:, None] + bias[:, None]
    return y


class PatchedF:
    """torch.nn.functional with conv1d intercepted for the layers the F(M,R) path would take (k in {7, 11}, >= 128 channels)."""

    def __init__(self, mode):
        self.mode, self.hits = mode, 0

    def __getattr__(self, name):
        return getattr(TF, name)

    def conv1d(self, x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        K = w.shape[2]
        if self.mode == "exact" or K not in (7, 11) or w.shape[1] < 128 or stride != 1 or groups != 1:
            return TF.conv1d(x, w, bias, stride, padding, dilation, groups)
        self.hits += 1
        assert padding == (K - 1) * dilation // 2
        if self.mode == "direct":
            return conv_direct(x, w, bias, dilation, padding)
        M = int(self.mode[2])
        return conv_toom(x, w, bias, dilation, padding, M, M)


def main():
    torch.manual_seed(0)
    man = manifest("ljspeech")
    dc = man["config"]["decoder"]
    dec = Decoder(**decoder_kwargs(dc)).eval()
    synth.init_synthetic_(dec, 1)
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    B, T = 1, 24
    asr, F0, N, s, noise = synth.decoder_inputs(B, T, 5)
    waves = {}
    for mode in ("exact", "direct", "F(3,3)", "F(4,4)", "F(6,6)"):
        f = PatchedF(mode)
        O.F = f
        try:
            with torch.no_grad():
                waves[mode] = O.decoder(sd, dc, asr, F0, N, s, noise=noise)
        finally:
            O.F = TF
        print("%-7s intercepted convs %3d   waveform abs-max %.3f rms %.4f" % (mode, f.hits, float(waves[mode].abs().max()), rms(waves[mode])))
    ref = waves["exact"]
    for mode in ("direct", "F(3,3)", "F(4,4)", "F(6,6)"):
        print("%-7s vs exact: waveform RMS error %.3e (bar 1e-4)   max |diff| %.3e" % (mode, rms(waves[mode] - ref),
                                                                                 float((waves[mode] - ref).abs().max())))


if __name__ == "__main__":
    main()
