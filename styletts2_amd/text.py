"""Text side of the path: TextEncoder (models.py:284-345), PL-BERT wrapper (Utils/PLBERT/util.py:6-12) and the
ProsodyPredictor with its DurationEncoder (models.py:440-582).

A module forward on a HIP tensor is ONE C-ABI call into the module's C++ launch plan (`st2_text_forward`,
`st2_bert_forward`, `st2_duration_forward`; csrc/st2_engine.hip) -- in `pipeline.inference` the whole front is one call
(`st2_front_forward`) and these module-level forms serve callers that follow the notebooks step by step.  Every conv,
every BiLSTM and the whole ALBERT encoder run on the HIP kernels: TextEncoder k=5 convs, F0/N AdainResBlk1d stacks and
every Linear (token-merged k=1 convs) on the split-f16 MFMA convs, LayerNorm / AdaLayerNorm + LeakyReLU + masking on
`st2_colnorm_stats` / `st2_colnorm_apply`, LSTM input projections as k=1 convs and the recurrences on
`st2_lstm_bidir_coop`, PL-BERT attention on `st2_attention_keylen`.  There is no PyTorch / HF forward behind any of them.
The per-kernel Python plans below (the bodies after the `_engine_path` test) are the tap-point path of the parity tests
(`_hooks.override(plan="python")`) and what the CPU plan tests step through; they issue the same kernels, with a few
PyTorch glue ops (embedding gathers, the style concatenation) where the C++ plans have kernels of their own.
State_dict layouts are the reference's, key for key.
"""
import weakref

import torch
import torch.nn as nn

from .layers import transient_state
from . import _hooks, ops
from . import weights as W
from .decoder import StyleBank, _PackedAdainResBlk, _PackedConv, run_adain_resblk
from .layers import AdainResBlk1dParams, PlainConv1d, PlainLinear, WNConv1d


def _engine_path(dev):
    """HIP tensor and the product's plan selector: the module forward is one call into its C++ launch plan."""
    return dev.type == "cuda" and _hooks.plan == "engine"


def _cached_engine(owner, attr, modules, build):
    """st2_engine handle packed once per (weights, device) and kept on `owner`: rebuilt when a parameter was reloaded
    (in-place version counter) or moved (storage address)."""
    stamp = tuple((p.data_ptr(), p._version) for m in modules for p in list(m.parameters()) + list(m.buffers()))
    c = owner.__dict__.get(attr)
    if c is None or c[0] != stamp:
        c = (stamp, build())
        owner.__dict__[attr] = c
    return c[1]


def _device_lengths(lengths, n, device):
    """int32 device copy of `lengths`, or None when no sequence is padded (decided on the host copy the caller
    already holds -- the reference reads `input_lengths.cpu()` at the same place, models.py:314)."""
    if lengths is None:
        return None
    if lengths.is_cuda:  # already the device copy of a batch the caller knows to be padded (pipeline.prepare)
        return lengths if lengths.dtype == torch.int32 else lengths.to(torch.int32)
    lc = lengths.detach().cpu()
    if bool((lc == n).all()):
        return None
    return lc.to(torch.int32).to(device)


@transient_state
class EngineLSTM(nn.LSTM):
    """nn.LSTM(1 layer, bidirectional, batch_first) parameter holder -- same state_dict keys as the reference's
    nn.LSTM -- whose arithmetic runs on the HIP kernels: the input projection of every time step is one k=1
    `st2_conv1d` (channel-major tokens), the recurrence is `st2_lstm_bidir`.  Pack/pad semantics of
    models.py:314-327 are reproduced through `lengths` (outputs past a sequence's end are zero)."""

    def __init__(self, input_size, hidden_size):
        super().__init__(input_size, hidden_size, 1, batch_first=True, bidirectional=True)
        self._pk = None

    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._pk = None
        return super().load_state_dict(state_dict, *a, **k)

    def _load_from_state_dict(self, *a, **k):  # also reached when a PARENT module's load_state_dict() recurses here
        self._pk = None
        return super()._load_from_state_dict(*a, **k)

    def _packed(self, device):
        if self._pk is None or self._pk.device != device:
            d = lambda t: t.detach().float().contiguous().to(device)
            pk = type("PackedLSTM", (), {})()
            pk.device = device
            w_ih = torch.cat([self.weight_ih_l0.detach(), self.weight_ih_l0_reverse.detach()], dim=0).float()
            pk.w_ih = W.pack_linear_auto(w_ih).to(device)                             # [I, 8H] (split-f16 by default)
            pk.bias = d(torch.cat([self.bias_ih_l0.detach() + self.bias_hh_l0.detach(),
                                   self.bias_ih_l0_reverse.detach() + self.bias_hh_l0_reverse.detach()]))
            pk.whh_t = d(torch.stack([self.weight_hh_l0.detach().t(), self.weight_hh_l0_reverse.detach().t()]))
            self._pk = pk
        return self._pk

    def forward_cm(self, x_cm, lengths=None):
        """x_cm [B, I, N] channel-major -> [B, 2H, N]; lengths: int32 device tensor or None."""
        pk = self._packed(x_cm.device)
        G = ops.conv1d(x_cm, pk.w_ih, 8 * self.hidden_size, 1, bias=pk.bias)
        return ops.lstm_bidir(G, pk.whh_t, lengths)

    @torch.no_grad()
    def forward(self, x, hx=None):
        """Reference call form `y, _ = lstm(x)` with x [B, N, I] (Demo/Inference_LJSpeech.ipynb:296)."""
        if hx is not None or not torch.is_tensor(x):
            raise NotImplementedError("EngineLSTM takes a padded [B, N, I] tensor; pass lengths via forward_cm")
        y = self.forward_cm(x.transpose(1, 2).contiguous().float())
        return y.transpose(1, 2), None


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

    def _load_from_state_dict(self, *a, **k):  # also reached when a PARENT module's load_state_dict() recurses here
        self._pk = None
        return super()._load_from_state_dict(*a, **k)

    def refresh(self):
        self._pk = None

    def _packed(self, device):
        pk = getattr(self, "_pk", None)
        if pk is None or pk.device != device:
            pk = self._prepare(device)
            pk.device = device
            self._pk = pk
        return pk


@transient_state
class TextEncoder(_PackedCache, nn.Module):
    """models.py:284-345: Embedding -> depth x [weight-norm Conv1d k5 -> LayerNorm(C) -> LeakyReLU(0.2)] -> BiLSTM."""

    def __init__(self, channels, kernel_size, depth, n_symbols):
        super().__init__()
        self.embedding = nn.Embedding(n_symbols, channels)
        self.kernel_size = kernel_size
        self.cnn = nn.ModuleList([nn.Sequential(WNConv1d(channels, channels, kernel_size), _ChannelLayerNorm(channels))
                                  for _ in range(depth)])
        self.lstm = EngineLSTM(channels, channels // 2)
        self._pk = None

    def _prepare(self, device):
        pk = type("PackedTextEncoder", (), {})()
        pk.convs = [_PackedConv(c[0], device) for c in self.cnn]
        return pk

    @torch.no_grad()
    def forward(self, x, input_lengths, m):
        """tokens [B, N], input_lengths [B], m = length_to_mask(input_lengths) -> t_en [B, C, N] (models.py:302-331; the
        mask is a function of the lengths in every reference call site, so the C++ plan takes the lengths)."""
        if _engine_path(x.device):
            from . import engine
            eng = _cached_engine(self, "_engine", [self], lambda: engine.build_text_engine(self, x.device))
            return eng.text_forward(x, _device_lengths(input_lengths, x.shape[1], x.device))
        pk = self._packed(x.device)
        h = self.embedding(x).transpose(1, 2).contiguous()  # [B, C, N]
        mk = m.to(x.device).unsqueeze(1)
        h.masked_fill_(mk, 0.0)
        lens = _device_lengths(input_lengths, h.shape[2], h.device)
        for c, pc in zip(self.cnn, pk.convs):
            h = ops.conv1d(h, pc.wt, pc.c_out, pc.ks, pad_left=(pc.ks - 1) // 2, bias=pc.bias)
            # LayerNorm over channels + LeakyReLU(0.2) + masked_fill in one pass (models.py:270-282,308-312)
            h = ops.colnorm_apply(h, ops.colnorm_stats(h, eps=c[1].eps), c[1].gamma.reshape(1, -1),
                                  c[1].beta.reshape(1, -1), act=ops.ACT_LEAKY, slope=0.2, lengths=lens)
        y = self.lstm.forward_cm(h, lens)  # [B, C, N]
        y.masked_fill_(mk, 0.0)
        return y


class _AdaLayerNorm(nn.Module):
    """models.py:418-438."""

    def __init__(self, style_dim, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.fc = PlainLinear(style_dim, channels * 2)

    def forward(self, x, s):
        raise NotImplementedError("AdaLayerNorm runs inside DurationEncoder.forward (st2_colnorm_stats / st2_colnorm_apply)")


@transient_state
class DurationEncoder(nn.Module):
    """models.py:517-569: nlayers x [BiLSTM(d_model+sty -> d_model), AdaLayerNorm, concat style]."""

    def __init__(self, sty_dim, d_model, nlayers, dropout=0.1):
        super().__init__()
        self.lstms = nn.ModuleList()
        for _ in range(nlayers):
            self.lstms.append(EngineLSTM(d_model + sty_dim, d_model // 2))
            self.lstms.append(_AdaLayerNorm(sty_dim, d_model))
        self.d_model, self.sty_dim = d_model, sty_dim

    @torch.no_grad()
    def forward(self, x, style, text_lengths, m):
        """x [B, d_model, N], style [B, sty] -> [B, N, d_model + sty] (models.py:536-569)."""
        B, _, N = x.shape
        lens = _device_lengths(text_lengths, N, x.device)
        owner = self.__dict__.get("_owner")
        owner = owner() if owner is not None else None
        if _engine_path(x.device) and owner is not None and owner.text_encoder is self:
            d_cm, _ = owner._pred_engine(x.device).duration_forward(x, style, lens, want_durations=False)
            return d_cm.transpose(1, 2)
        mk = m.to(x.device).unsqueeze(1)  # [B, 1, N]
        # channel-major throughout: [x | style] rows, the LSTMs and the AdaLayerNorm all work on [B, C, N]
        sty = style.float().unsqueeze(-1).expand(-1, -1, N)
        h = torch.cat([x.float(), sty], dim=1)
        h.masked_fill_(mk, 0.0)
        for block in self.lstms:
            if isinstance(block, _AdaLayerNorm):
                # [B, 2C]: gamma | beta (models.py:430-435) on st2_style_fc
                gb = ops.style_fc(style.float().contiguous(), block.fc.weight.detach().float().t().contiguous(),
                                  block.fc.bias.detach().float().contiguous())
                C = block.channels
                nh = torch.empty((B, C + self.sty_dim, N), device=h.device, dtype=torch.float32)
                ops.colnorm_apply(h, ops.colnorm_stats(h, eps=block.eps), gb[:, :C], gb[:, C:], gamma_plus_one=True,
                                  lengths=lens, out=nh[:, :C])
                nh[:, C:].copy_(sty)
                if lens is not None:
                    nh[:, C:].masked_fill_(mk, 0.0)
                h = nh
            else:
                h = block.forward_cm(h.contiguous(), lens)
        return h.transpose(1, 2)


@transient_state
class EngineLinear(_PackedCache, nn.Linear):
    """nn.Linear parameter holder (same state_dict keys) whose forward is a k=1 split-f16 MFMA conv over the merged
    leading dimensions (`bert_encoder`, models.py:689; `duration_proj.linear_layer`, models.py:34-44)."""

    def __init__(self, in_features, out_features):
        super().__init__(in_features, out_features)
        self._pk = None

    def _prepare(self, device):
        pk = type("PackedLinear", (), {})()
        pk.w = W.pack_linear_auto(self.weight.detach().float().cpu()).to(device)
        pk.b = self.bias.detach().float().contiguous().to(device)
        return pk

    @torch.no_grad()
    def forward(self, x):
        pk = self._packed(x.device)
        lead = x.shape[:-1]
        xc = x.reshape(1, -1, self.in_features).transpose(1, 2).contiguous().float()  # [1, in, M] channel-major tokens
        y = ops.conv1d(xc, pk.w, self.out_features, 1, bias=pk.b)                      # [1, out, M]
        return y.transpose(1, 2).reshape(*lead, self.out_features)


class _LinearNorm(nn.Module):
    """models.py:34-44 (`linear_layer` key)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.linear_layer = EngineLinear(in_dim, out_dim)

    def forward(self, x):
        return self.linear_layer(x)


@transient_state
class ProsodyPredictor(_PackedCache, nn.Module):
    """models.py:440-515.  `text_encoder`, `lstm`, `duration_proj`, `F0Ntrain` are called individually by the
    inference glue (Demo/Inference_LJSpeech.ipynb:294-311), so they keep the reference signatures."""

    def __init__(self, style_dim, d_hid, nlayers, max_dur=50, dropout=0.1):
        super().__init__()
        self.text_encoder = DurationEncoder(sty_dim=style_dim, d_model=d_hid, nlayers=nlayers, dropout=dropout)
        self.lstm = EngineLSTM(d_hid + style_dim, d_hid // 2)
        self.duration_proj = _LinearNorm(d_hid, max_dur)
        self.shared = EngineLSTM(d_hid + style_dim, d_hid // 2)
        mk = lambda: nn.ModuleList([AdainResBlk1dParams(d_hid, d_hid, style_dim),
                                    AdainResBlk1dParams(d_hid, d_hid // 2, style_dim, upsample=True),
                                    AdainResBlk1dParams(d_hid // 2, d_hid // 2, style_dim)])
        self.F0, self.N = mk(), mk()
        self.F0_proj = PlainConv1d(d_hid // 2, 1, 1)
        self.N_proj = PlainConv1d(d_hid // 2, 1, 1)
        self._pk = None
        self.text_encoder.__dict__["_owner"] = weakref.ref(self)  # its C++ plan lives in this module's engine handle

    def __setstate__(self, state):  # a copy owns its own text_encoder: point its plan at THIS predictor
        super().__setstate__(state)
        self.text_encoder.__dict__["_owner"] = weakref.ref(self)

    def _pred_engine(self, device):
        """The st2_engine handle holding the whole predictor (st2_duration_forward, st2_prosody_forward)."""
        from . import engine
        return _cached_engine(self, "_engine", [self], lambda: engine.build_predictor_engine(self, device))

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
        y = self.shared.forward_cm(x.float().contiguous())  # [B, d_hid, T]
        h = pk.bank.run(s.float())
        outs = []
        for blocks, (pw, pb) in ((pk.F0, (pk.f0p_w, pk.f0p_b)), (pk.N, (pk.np_w, pk.np_b))):
            t = y
            for blk in blocks:
                t = run_adain_resblk(blk, pk.bank, h, t)
            outs.append(ops.conv1d_direct(t, pw, pb, 1, 0).squeeze(1))
        return outs[0], outs[1]


def build_plbert(plbert_params):
    """PL-BERT (Utils/PLBERT/util.py:6-20): an HF AlbertModel subclass -- so `config`, the parameters and the
    state_dict are the reference's key for key -- whose forward runs on the engine's HIP kernels and returns
    `last_hidden_state`.  `transformers` is imported lazily so that the rest of the engine imports without it.

    Forward: one `st2_bert_forward` call (C++ plan); the per-kernel plan `forward_engine` below is the tests' tap path.
    In both, tokens are channel-major and token-merged ([768, B*N] storage), every Linear is one k=1 split-f16 MFMA conv over B*N columns (q|k|v fused into
    one 768->2304 conv), attention is `st2_attention_keylen` (12 heads x 64, key padding from the attention mask),
    the post-LN residual blocks are `st2_colnorm_stats` + `st2_colnorm_apply` (eps 1e-12), the FFN activation
    (gelu_new) sits in the conv epilogue; the embedding LayerNorm is the prologue of the 128->768 mapping conv.
    The 12 layers share one weight set (ALBERT), packed once per load."""
    from transformers import AlbertConfig, AlbertModel

    @transient_state
    class CustomAlbert(AlbertModel):
        def _apply(self, fn, *a, **k):
            self._pk = None
            return super()._apply(fn, *a, **k)

        def load_state_dict(self, state_dict, *a, **k):
            self._pk = None
            return super().load_state_dict(state_dict, *a, **k)

        def _load_from_state_dict(self, *a, **k):
            self._pk = None
            return super()._load_from_state_dict(*a, **k)

        def refresh(self):
            self._pk = None

        def _packed(self, device):
            pk = getattr(self, "_pk", None)
            if pk is not None and pk.device == device:
                return pk
            cfg = self.config
            assert cfg.num_hidden_groups == 1 and cfg.inner_group_num == 1, "PL-BERT shares one ALBERT layer"
            assert cfg.hidden_act == "gelu_new" and cfg.hidden_size // cfg.num_attention_heads == 64
            d = lambda t: t.detach().float().contiguous().to(device)
            row = lambda t: d(t).reshape(1, -1)
            lay = self.encoder.albert_layer_groups[0].albert_layers[0]
            att = lay.attention
            pk = type("PackedAlbert", (), {})()
            pk.device = device
            emb = self.embeddings
            pk.word, pk.pos, pk.tok0 = d(emb.word_embeddings.weight), d(emb.position_embeddings.weight), d(
                emb.token_type_embeddings.weight[0])
            pk.eln_w, pk.eln_b = row(emb.LayerNorm.weight), row(emb.LayerNorm.bias)
            pk.map, pk.map_b = (W.pack_linear_auto(self.encoder.embedding_hidden_mapping_in.weight.detach().float())
                                .to(device), d(self.encoder.embedding_hidden_mapping_in.bias))
            wqkv = torch.cat([att.query.weight, att.key.weight, att.value.weight], dim=0).detach().float()
            pk.qkv, pk.qkv_b = W.pack_linear_auto(wqkv).to(device), d(torch.cat([att.query.bias, att.key.bias,
                                                                               att.value.bias]))
            pk.dense, pk.dense_b = W.pack_linear_auto(att.dense.weight.detach().float()).to(device), d(att.dense.bias)
            pk.aln_w, pk.aln_b = row(att.LayerNorm.weight), row(att.LayerNorm.bias)
            pk.ffn, pk.ffn_b = W.pack_linear_auto(lay.ffn.weight.detach().float()).to(device), d(lay.ffn.bias)
            pk.out, pk.out_b = W.pack_linear_auto(lay.ffn_output.weight.detach().float()).to(device), d(
                lay.ffn_output.bias)
            pk.fln_w, pk.fln_b = row(lay.full_layer_layer_norm.weight), row(lay.full_layer_layer_norm.bias)
            self._pk = pk
            return pk

        @torch.no_grad()
        def forward(self, input_ids=None, attention_mask=None, **kwargs):
            """`bert(tokens, attention_mask=(~text_mask).int())` (Demo/Inference_LJSpeech.ipynb:284) -> last hidden state
            [B, N, hidden].  Only this call form exists: there is no HF forward behind it, other HF options raise."""
            if kwargs:
                raise TypeError("PL-BERT on the MI355X engine takes (input_ids, attention_mask) only; got %s"
                                % sorted(kwargs))
            if _engine_path(input_ids.device):
                from . import engine
                eng = _cached_engine(self, "_engine", [self], lambda: engine.build_bert_engine(self, input_ids.device))
                lens = None
                if attention_mask is not None:  # right-padded batch (length_to_mask): valid keys are a prefix
                    lens = attention_mask.to(torch.int32).sum(dim=1).to(torch.int32).contiguous()
                return eng.bert_forward(input_ids, lens)
            return self.forward_engine(input_ids, attention_mask)  # no CPU fallback: non-HIP tensors raise in ops

        @torch.no_grad()
        def forward_engine(self, input_ids, attention_mask=None):
            cfg = self.config
            pk = self._packed(input_ids.device)
            B, N = input_ids.shape
            Hd, heads, eps = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps
            dev = input_ids.device
            key_len = None
            if attention_mask is not None:  # right-padded batch (length_to_mask): valid keys are a prefix
                key_len = attention_mask.to(torch.int32).sum(dim=1).to(torch.int32).contiguous()

            def alloc(C):  # [B, C, N] view of a [C, B*N] block (token-merged channel-major)
                return torch.empty((C, B, N), device=dev, dtype=torch.float32).permute(1, 0, 2)

            cv = lambda t: t.permute(1, 0, 2).reshape(1, t.shape[1], B * N)  # the k=1 convs' view (no copy)
            E = alloc(cfg.embedding_size)
            E.copy_((pk.word[input_ids] + pk.tok0 + pk.pos[:N].unsqueeze(0)).permute(0, 2, 1))  # gather glue
            st = ops.colnorm_stats(E, eps=eps)
            X = alloc(Hd)
            ops.conv1d(cv(E), pk.map, Hd, 1, bias=pk.map_b, pro=ops.PRO_COLNORM, stats=st.view(1, B * N, 2),
                       gamma=pk.eln_w, beta=pk.eln_b, out=cv(X))
            for _ in range(cfg.num_hidden_layers):
                qkv = alloc(3 * Hd)
                ops.conv1d(cv(X), pk.qkv, 3 * Hd, 1, bias=pk.qkv_b, out=cv(qkv))
                ctx = ops.attention(qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:], heads, 64 ** -0.5, out=alloc(Hd),
                                    key_len=key_len)
                Y = alloc(Hd)
                ops.conv1d(cv(ctx), pk.dense, Hd, 1, bias=pk.dense_b, res=cv(X), out=cv(Y))
                X1 = ops.colnorm_apply(Y, ops.colnorm_stats(Y, eps=eps), pk.aln_w, pk.aln_b, out=alloc(Hd))
                Hm = alloc(cfg.intermediate_size)
                ops.conv1d(cv(X1), pk.ffn, cfg.intermediate_size, 1, bias=pk.ffn_b, act=ops.ACT_GELU_TANH, out=cv(Hm))
                Z = alloc(Hd)
                ops.conv1d(cv(Hm), pk.out, Hd, 1, bias=pk.out_b, res=cv(X1), out=cv(Z))
                X = ops.colnorm_apply(Z, ops.colnorm_stats(Z, eps=eps), pk.fln_w, pk.fln_b, out=alloc(Hd))
            return X.permute(0, 2, 1)  # [B, N, 768] view

    return CustomAlbert(AlbertConfig(**plbert_params))
