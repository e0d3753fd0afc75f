"""Text side of the path: TextEncoder (models.py:284-345), PL-BERT wrapper (Utils/PLBERT/util.py:6-12) and the
ProsodyPredictor with its DurationEncoder (models.py:440-582).

Scope note: the recurrent cells (BiLSTM, H=256) and the ALBERT encoder run through
PyTorch-ROCm (MIOpen RNN / hipBLASLt) this round -- they are ~7 % of the path's time.  Everything conv-shaped
here (TextEncoder k=5 convs, the F0/N AdainResBlk1d stacks) already runs on the HIP kernels.  State_dict layouts
are the reference's, key for key.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from . import weights as W
from .decoder import StyleBank, _PackedAdainResBlk, _PackedConv, run_adain_resblk
from .layers import AdainResBlk1dParams, PlainConv1d, PlainLinear, WNConv1d


def _all_full(lengths, n):
    """True when no sequence is padded; decided on the host copy of `lengths` (kept on CPU by the caller)."""
    return bool((lengths == n).all())


def _bilstm(lstm, x, lengths, total):
    """Batch-first BiLSTM with the reference's pack/pad semantics (models.py:314-327): padded steps are skipped
    by the recurrence and come out as zeros."""
    lstm.flatten_parameters()
    lens_cpu = lengths.detach().cpu()
    if _all_full(lens_cpu, x.shape[1]):
        y, _ = lstm(x)
        return y
    packed = nn.utils.rnn.pack_padded_sequence(x, lens_cpu, batch_first=True, enforce_sorted=False)
    y, _ = lstm(packed)
    y, _ = nn.utils.rnn.pad_packed_sequence(y, batch_first=True, total_length=total)
    return y


class _ChannelLayerNorm(nn.Module):
    """models.py:270-282 (`gamma`/`beta` parameter names)."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))


class _PackedCache:
    """Shared lazy packed-weight cache (invalidated by .to()/load_state_dict())."""

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._pk = None
        return super().load_state_dict(W.strip_module_prefix(state_dict), *a, **k)

    def refresh(self):
        self._pk = None

    def _packed(self, device):
        pk = getattr(self, "_pk", None)
        if pk is None or pk.device != device:
            pk = self._prepare(device)
            pk.device = device
            self._pk = pk
        return pk


class TextEncoder(_PackedCache, nn.Module):
    """models.py:284-345: Embedding -> depth x [weight-norm Conv1d k5 -> LayerNorm(C) -> LeakyReLU(0.2)] -> BiLSTM."""

    def __init__(self, channels, kernel_size, depth, n_symbols):
        super().__init__()
        self.embedding = nn.Embedding(n_symbols, channels)
        self.kernel_size = kernel_size
        self.cnn = nn.ModuleList([nn.Sequential(WNConv1d(channels, channels, kernel_size), _ChannelLayerNorm(channels))
                                  for _ in range(depth)])
        self.lstm = nn.LSTM(channels, channels // 2, 1, batch_first=True, bidirectional=True)
        self._pk = None

    def _prepare(self, device):
        pk = type("PackedTextEncoder", (), {})()
        pk.convs = [_PackedConv(c[0], device) for c in self.cnn]
        return pk

    @torch.no_grad()
    def forward(self, x, input_lengths, m):
        pk = self._packed(x.device)
        h = self.embedding(x).transpose(1, 2).contiguous()  # [B, C, N]
        mk = m.to(x.device).unsqueeze(1)
        h.masked_fill_(mk, 0.0)
        for c, pc in zip(self.cnn, pk.convs):
            h = ops.conv1d(h, pc.wt, pc.c_out, pc.ks, pad_left=(pc.ks - 1) // 2, bias=pc.bias)
            h = F.layer_norm(h.transpose(1, 2), (c[1].channels,), c[1].gamma, c[1].beta, c[1].eps).transpose(1, 2)
            h = F.leaky_relu(h, 0.2).contiguous()
            h.masked_fill_(mk, 0.0)
        y = _bilstm(self.lstm, h.transpose(1, 2), input_lengths, m.shape[-1])
        y = y.transpose(-1, -2).contiguous()
        y.masked_fill_(mk, 0.0)
        return y


class _AdaLayerNorm(nn.Module):
    """models.py:418-438."""

    def __init__(self, style_dim, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.fc = PlainLinear(style_dim, channels * 2)

    def forward(self, x, s):  # x [B, N, C]
        h = F.linear(s, self.fc.weight, self.fc.bias)
        gamma, beta = torch.chunk(h.unsqueeze(1), 2, dim=-1)
        return (1 + gamma) * F.layer_norm(x, (self.channels,), eps=self.eps) + beta


class DurationEncoder(nn.Module):
    """models.py:517-569: nlayers x [BiLSTM(d_model+sty -> d_model), AdaLayerNorm, concat style]."""

    def __init__(self, sty_dim, d_model, nlayers, dropout=0.1):
        super().__init__()
        self.lstms = nn.ModuleList()
        for _ in range(nlayers):
            self.lstms.append(nn.LSTM(d_model + sty_dim, d_model // 2, num_layers=1, batch_first=True,
                                      bidirectional=True))
            self.lstms.append(_AdaLayerNorm(sty_dim, d_model))
        self.d_model, self.sty_dim = d_model, sty_dim

    @torch.no_grad()
    def forward(self, x, style, text_lengths, m):
        """x [B, d_model, N], style [B, sty] -> [B, N, d_model + sty]."""
        mk = m.to(x.device)
        N = x.shape[2]
        s = style.unsqueeze(1).expand(-1, N, -1)
        h = torch.cat([x.transpose(1, 2), s], dim=-1)
        h = h.masked_fill(mk.unsqueeze(-1), 0.0)
        for block in self.lstms:
            if isinstance(block, _AdaLayerNorm):
                h = block(h, style)
                h = torch.cat([h, s], dim=-1)
                h = h.masked_fill(mk.unsqueeze(-1), 0.0)
            else:
                h = _bilstm(block, h, text_lengths, m.shape[-1])
        return h


class _LinearNorm(nn.Module):
    """models.py:34-44 (`linear_layer` key)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim)

    def forward(self, x):
        return self.linear_layer(x)


class ProsodyPredictor(_PackedCache, nn.Module):
    """models.py:440-515.  `text_encoder`, `lstm`, `duration_proj`, `F0Ntrain` are called individually by the
    inference glue (Demo/Inference_LJSpeech.ipynb:294-311), so they keep the reference signatures."""

    def __init__(self, style_dim, d_hid, nlayers, max_dur=50, dropout=0.1):
        super().__init__()
        self.text_encoder = DurationEncoder(sty_dim=style_dim, d_model=d_hid, nlayers=nlayers, dropout=dropout)
        self.lstm = nn.LSTM(d_hid + style_dim, d_hid // 2, 1, batch_first=True, bidirectional=True)
        self.duration_proj = _LinearNorm(d_hid, max_dur)
        self.shared = nn.LSTM(d_hid + style_dim, d_hid // 2, 1, batch_first=True, bidirectional=True)
        mk = lambda: nn.ModuleList([AdainResBlk1dParams(d_hid, d_hid, style_dim),
                                    AdainResBlk1dParams(d_hid, d_hid // 2, style_dim, upsample=True),
                                    AdainResBlk1dParams(d_hid // 2, d_hid // 2, style_dim)])
        self.F0, self.N = mk(), mk()
        self.F0_proj = PlainConv1d(d_hid // 2, 1, 1)
        self.N_proj = PlainConv1d(d_hid // 2, 1, 1)
        self._pk = None

    def _prepare(self, device):
        pk = type("PackedPredictor", (), {})()
        bank = StyleBank()
        for blk in list(self.F0) + list(self.N):
            bank.add(blk.norm1)
            bank.add(blk.norm2)
        bank.pack(device)
        pk.bank = bank
        pk.F0 = [_PackedAdainResBlk(b, device) for b in self.F0]
        pk.N = [_PackedAdainResBlk(b, device) for b in self.N]
        d = lambda t: t.detach().float().contiguous().to(device)
        pk.f0p_w, pk.f0p_b = d(self.F0_proj.weight), d(self.F0_proj.bias)
        pk.np_w, pk.np_b = d(self.N_proj.weight), d(self.N_proj.bias)
        return pk

    @torch.no_grad()
    def F0Ntrain(self, x, s):
        """x [B, d_hid+sty, T] -> (F0 [B, 2T], N [B, 2T]); models.py:497-510."""
        pk = self._packed(x.device)
        self.shared.flatten_parameters()
        y, _ = self.shared(x.transpose(-1, -2))
        y = y.transpose(-1, -2).contiguous()
        h = pk.bank.run(s.float())
        outs = []
        for blocks, (pw, pb) in ((pk.F0, (pk.f0p_w, pk.f0p_b)), (pk.N, (pk.np_w, pk.np_b))):
            t = y
            for blk in blocks:
                t = run_adain_resblk(blk, pk.bank, h, t)
            outs.append(ops.conv1d_direct(t, pw, pb, 1, 0).squeeze(1))
        return outs[0], outs[1]


def build_plbert(plbert_params):
    """PL-BERT (Utils/PLBERT/util.py:6-20): an HF AlbertModel subclass whose forward returns
    `last_hidden_state`, so its state_dict is the reference's key for key.  `transformers` is imported lazily so
    that the rest of the engine imports without it."""
    from transformers import AlbertConfig, AlbertModel

    class CustomAlbert(AlbertModel):
        @torch.no_grad()
        def forward(self, *args, **kwargs):
            return super().forward(*args, **kwargs).last_hidden_state

    return CustomAlbert(AlbertConfig(**plbert_params))
