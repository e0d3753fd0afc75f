"""Style-diffusion sampler engine: drop-in for the reference classes used at inference
(Modules/diffusion/sampler.py: KarrasSchedule :319-337, ADPM2Sampler :481-519, KDiffusion :165-208,
DiffusionSampler :550-586; Modules/diffusion/modules.py: Transformer1d :283-427, StyleTransformer1d :40-185;
Modules/diffusion/diffusion.py: AudioDiffusionConditional shell, re-wired by models.py:653-669).

    sampler = DiffusionSampler(model.diffusion.diffusion, sampler=ADPM2Sampler(),
                               sigma_schedule=KarrasSchedule(sigma_min=1e-4, sigma_max=3.0, rho=9.0), clamp=False)
    s_pred = sampler(noise[B,1,256], embedding=bert_dur[B,N,768], embedding_scale=1.0, num_steps=5[, features=ref_s])

MI355X design: tokens are kept CHANNEL-MAJOR ([B, C, N]) for the whole denoiser so that every Linear is a k=1
conv on the matrix pipe (`st2_conv1d_f16s`: split-f16 MFMA, fp32-class accuracy; `st2_conv1d` = the exact-fp32 build of the same
contract, which the parity tests select through `_hooks.override(conv_precision="f32")`) with LayerNorm/AdaLayerNorm applied in the conv prologue, GELU / residual in its epilogue;
attention is one fp32-MFMA HIP kernel; the per-utterance mapping MLP is `st2_style_fc`.  Everything the ADPM2
loop needs from the host (sigma schedule, sigma_up/down/mid, EDM scale weights) is input-independent and is
computed once on the host in the reference's own arithmetic (fp32 tensors + python floats), so the loop issues
kernels back to back with no device->host synchronisation (the reference syncs every step, sampler.py:490-495).
"""
import math

import torch
import torch.nn as nn

from .layers import transient_state
from . import ops
from . import weights as W
from .layers import PlainConv1d, PlainLinear


# ---------------------------------------------------------------------------------------------------
# schedule / sampler / diffusion wrappers (host-side, reference API)
# ---------------------------------------------------------------------------------------------------
class KarrasSchedule(nn.Module):
    """sampler.py:319-337."""

    def __init__(self, sigma_min: float, sigma_max: float, rho: float = 7.0):
        super().__init__()
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def forward(self, num_steps: int, device=None):
        rho_inv = 1.0 / self.rho
        steps = torch.arange(num_steps, dtype=torch.float32)  # host: the schedule is input independent
        sigmas = (self.sigma_max ** rho_inv +
                  (steps / (num_steps - 1)) * (self.sigma_min ** rho_inv - self.sigma_max ** rho_inv)) ** self.rho
        return torch.nn.functional.pad(sigmas, pad=(0, 1), value=0.0)


class ADPM2Sampler(nn.Module):
    """sampler.py:481-519 (second-order ancestral DPM step)."""

    def __init__(self, rho: float = 1.0):
        super().__init__()
        self.rho = rho

    def get_sigmas(self, sigma, sigma_next):
        """Same expressions on the same types as the reference (0-dim fp32 tensors -> python floats)."""
        r = self.rho
        sigma_up = math.sqrt(sigma_next ** 2 * (sigma ** 2 - sigma_next ** 2) / sigma ** 2)
        sigma_down = math.sqrt(sigma_next ** 2 - sigma_up ** 2)
        sigma_mid = ((sigma ** (1 / r) + sigma_down ** (1 / r)) / 2) ** r
        return float(sigma_up), float(sigma_down), float(sigma_mid)


class KDiffusion(nn.Module):
    """EDM pre-conditioning around the denoiser, sampler.py:165-208.  `sigma_data` is a plain mutable float
    (models.py:665, train_second.py:317), not part of the state_dict."""

    alias = "k"

    def __init__(self, net, *, sigma_data: float, sigma_distribution=None, dynamic_threshold: float = 0.0):
        super().__init__()
        self.net = net
        self.sigma_data = sigma_data
        self.sigma_distribution = sigma_distribution
        self.dynamic_threshold = dynamic_threshold

    def get_scale_weights(self, sigma: float):
        """sampler.py:184-191, evaluated in fp32 like the reference's tensor arithmetic."""
        sd = self.sigma_data
        s = torch.tensor(float(sigma), dtype=torch.float32)
        c_noise = torch.log(s) * 0.25
        c_skip = (sd ** 2) / (s ** 2 + sd ** 2)
        c_out = s * sd * (sd ** 2 + s ** 2) ** -0.5
        c_in = (s ** 2 + sd ** 2) ** -0.5
        return float(c_skip), float(c_out), float(c_in), float(c_noise)

    @torch.no_grad()
    def denoise_fn(self, x_noisy, sigmas=None, sigma=None, _session=None, **kwargs):
        if sigma is None:
            if sigmas is None:
                raise ValueError("denoise_fn needs sigma or sigmas")
            sv = sigmas.reshape(-1)
            if sv.numel() > 1 and not bool((sv == sv[0]).all()):
                raise NotImplementedError("per-item sigmas are a training-time feature (sampler.py:223)")
            sigma = float(sv[0])
        c_skip, c_out, c_in, c_noise = self.get_scale_weights(float(sigma))
        sess = _session if _session is not None else self.net.open_session(x_noisy, **kwargs)
        x_in = ops.axpbypcz(x_noisy.contiguous(), c_in)
        x_pred = self.net.run_session(sess, x_in, c_noise)
        return ops.axpbypcz(x_noisy.contiguous(), c_skip, x_pred, c_out)


@transient_state
class DiffusionSampler(nn.Module):
    """sampler.py:550-586."""

    def __init__(self, diffusion, *, sampler, sigma_schedule, num_steps=None, clamp=True):
        super().__init__()
        assert getattr(diffusion, "alias", None) == "k", "ADPM2Sampler incompatible with %s" % type(diffusion).__name__
        self.diffusion = diffusion
        self.denoise_fn = diffusion.denoise_fn
        self.sampler = sampler
        self.sigma_schedule = sigma_schedule
        self.num_steps = num_steps
        self.clamp = clamp
        self._tables = {}

    def step_table(self, num_steps):
        """The input-independent scalars of the ADPM2 loop in the reference's own arithmetic (fp32 tensors for the
        schedule and the EDM weights, python floats for sigma_up / down / mid): `st2_sampler_run`'s `table` (11 doubles per
        step, include/st2.h) and sigma0."""
        cached = self._tables.get(num_steps)  # input independent: computed once per step count
        if cached is not None:
            return cached
        sigmas = self.sigma_schedule(num_steps, None)
        table = []
        for i in range(num_steps - 1):
            sigma, sigma_next = sigmas[i], sigmas[i + 1]
            s_up, s_down, s_mid = self.sampler.get_sigmas(sigma, sigma_next)
            sg = float(sigma)
            table += list(self.diffusion.get_scale_weights(sg)) + list(self.diffusion.get_scale_weights(s_mid))
            table += [(s_mid - sg) / sg, (s_down - sg) / s_mid, s_up]
        self._tables[num_steps] = (table, float(sigmas[0]))
        return self._tables[num_steps]

    def _engine(self, device):
        """The C++ plan's handle for this denoiser on `device` (packed once per load; rebuilt when the Python-side
        packed-weight cache was invalidated by load_state_dict / .to())."""
        from . import engine
        net = self.diffusion.net
        eng = getattr(net, "_engine", None)
        if eng is None or eng.device != device or getattr(net, "_engine_stale", True):
            eng = engine.build_denoiser_engine(net, device)
            net._engine, net._engine_stale = eng, False
        return eng

    @torch.no_grad()
    def forward(self, noise, num_steps=None, step_noise=None, taps=None, **kwargs):
        """`step_noise` [num_steps-1, B, 1, C] (optional) replays the per-step randn_like draws of sampler.py:509."""
        num_steps = num_steps if num_steps is not None else self.num_steps
        assert num_steps is not None, "Parameter `num_steps` must be provided"
        from . import engine
        if noise.is_cuda and engine.plan_mode() == "engine" and not kwargs.get("embedding_mask_proba"):
            return self._forward_engine(noise, num_steps, step_noise, taps, **kwargs)
        sigmas = self.sigma_schedule(num_steps, noise.device)
        noise = noise.float().contiguous()
        sess = self.diffusion.net.open_session(noise, **kwargs)
        fn = lambda xx, sg: self.diffusion.denoise_fn(xx, sigma=sg, _session=sess)
        x = ops.axpbypcz(noise, float(sigmas[0]))
        for i in range(num_steps - 1):
            sigma, sigma_next = sigmas[i], sigmas[i + 1]
            s_up, s_down, s_mid = self.sampler.get_sigmas(sigma, sigma_next)
            sg = float(sigma)
            # d = (x - fn(x, sigma)) / sigma ; x_mid = x + d * (sigma_mid - sigma)
            k = (s_mid - sg) / sg
            x_mid = ops.axpbypcz(x, 1.0 + k, fn(x, sg), -k)
            # d_mid = (x_mid - fn(x_mid, sigma_mid)) / sigma_mid ; x = x + d_mid * (sigma_down - sigma) + eps * s_up
            k2 = (s_down - sg) / s_mid
            eps = step_noise[i].float().contiguous() if step_noise is not None else torch.randn_like(x)
            den_mid = fn(x_mid, s_mid)
            d_mid = ops.axpbypcz(x_mid, k2, den_mid, -k2)
            x = ops.axpbypcz(x, 1.0, d_mid, 1.0, eps, s_up)
            if taps is not None:
                taps["step%d" % i] = x
        return x.clamp(-1.0, 1.0) if self.clamp else x

    def _forward_engine(self, noise, num_steps, step_noise, taps, embedding=None, features=None, embedding_scale=1.0,
                        lengths=None, embedding_mask_proba=0.0):
        """One `st2_sampler_run` call: the whole ADPM2 loop is issued by the C++ plan (csrc/st2_engine.hip)."""
        assert embedding is not None, "the denoiser is conditional: `embedding` is required (modules.py:410)"
        net = self.diffusion.net
        B, N, E = embedding.shape
        assert N <= net.fixed_embedding.max_length, "Input sequence length must be <= max_length"
        if net.multispeaker:
            assert features is not None, "context_features exists but no features provided"
        if step_noise is None:  # the reference's per-step randn_like draws (sampler.py:509)
            step_noise = torch.randn((num_steps - 1,) + tuple(noise.shape), device=noise.device, dtype=torch.float32)
        if lengths is not None and not (lengths.dtype == torch.int32 and lengths.device == noise.device):
            assert int(lengths.min()) >= 1 and int(lengths.max()) <= N
            lengths = lengths.to(torch.int32).to(noise.device)
        table, sigma0 = self.step_table(num_steps)
        x = self._engine(noise.device).sampler_run(noise, embedding, features if net.multispeaker else None, step_noise,
                                                   None if lengths is None else lengths.contiguous(), num_steps,
                                                   float(embedding_scale), table, sigma0, taps=taps)
        return x.clamp(-1.0, 1.0) if self.clamp else x


class GraphedSampler(nn.Module):
    """hipGraph replay of a whole `DiffusionSampler` run (BASELINE.json configs[4]: "hipGraph-captured diffusion
    step").  One sampler run is ~300 launches of 10-40 us kernels whose sequence depends only on (batch, tokens, steps,
    guidance scale): the first call with a new signature runs eagerly once (packs weights, sets kernel attributes),
    then records the run into a `torch.cuda.CUDAGraph` (a hipGraph on ROCm) with static input / output buffers; later
    calls copy their inputs in, replay the graph and return a copy of the output -- one graph launch instead of ~300
    kernel launches, so a latency-bound caller (sentence-by-sentence long-form synthesis) is no longer paced by the
    host.  Same call signature and results as the wrapped sampler; the per-step noise is always an explicit input
    (drawn here with torch.randn when the caller gives none) so that replays do not repeat a captured draw."""

    def __init__(self, sampler, max_graphs=16):
        super().__init__()
        self.sampler = sampler
        self.max_graphs = max_graphs
        self._graphs = {}

    @torch.no_grad()
    def forward(self, noise, num_steps=None, step_noise=None, embedding=None, features=None, embedding_scale=1.0,
                taps=None, lengths=None, **kwargs):
        num_steps = num_steps if num_steps is not None else self.sampler.num_steps
        if (not noise.is_cuda) or taps is not None or kwargs or embedding is None:
            if lengths is not None:
                kwargs["lengths"] = lengths
            return self.sampler(noise, num_steps=num_steps, step_noise=step_noise, embedding=embedding,
                                features=features, embedding_scale=embedding_scale, taps=taps, **kwargs)
        B, N = embedding.shape[0], embedding.shape[1]
        if step_noise is None:
            step_noise = torch.randn((num_steps - 1,) + tuple(noise.shape), device=noise.device, dtype=torch.float32)
        key = (noise.device.index, B, N, int(num_steps), float(embedding_scale), features is not None,
               lengths is not None)
        g = self._graphs.get(key)
        net = self.sampler.diffusion.net
        if g is not None and g["gen"] != getattr(net, "_pack_gen", 0):
            # the denoiser's packed weights were rebuilt (load_state_dict / .to()): every recorded graph still points
            # at the old, freed pack
            self._graphs.clear()
            g = None
        if g is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            g = self._capture(noise, num_steps, step_noise, embedding, features, embedding_scale, lengths)
            self._graphs[key] = g
        if lengths is not None:
            g["lengths"].copy_(lengths)
        g["noise"].copy_(noise)
        g["step_noise"].copy_(step_noise)
        g["embedding"].copy_(embedding)
        if features is not None:
            g["features"].copy_(features)
        g["graph"].replay()
        return g["out"].clone()

    def _capture(self, noise, num_steps, step_noise, embedding, features, embedding_scale, lengths=None):
        st = dict(noise=noise.detach().float().clone(), step_noise=step_noise.detach().float().clone(),
                  embedding=embedding.detach().float().clone(),
                  features=None if features is None else features.detach().float().clone(),
                  lengths=None if lengths is None else lengths.detach().to(torch.int32).to(noise.device).clone())

        def run():
            kw = dict(num_steps=num_steps, step_noise=st["step_noise"], embedding=st["embedding"],
                      embedding_scale=embedding_scale)
            if st["features"] is not None:
                kw["features"] = st["features"]
            if st["lengths"] is not None:
                kw["lengths"] = st["lengths"]  # int32 on the device: read by the kernels, never by the host
            return self.sampler(st["noise"], **kw)

        cur = torch.cuda.current_stream(noise.device)
        side = ops.aux_stream(noise.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            run()  # eager warm-up on the capture stream: weight packing, hipFuncSetAttribute, allocator warm-up
        cur.wait_stream(side)
        torch.cuda.synchronize(noise.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            st["out"] = run()
        st["graph"] = graph
        st["gen"] = getattr(self.sampler.diffusion.net, "_pack_gen", 0)  # generation of the packed weights the graph reads
        return st


# ---------------------------------------------------------------------------------------------------
# denoiser parameter holders (state_dict layout of modules.py) + engine
# ---------------------------------------------------------------------------------------------------
class _LearnedPositionalEmbedding(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(dim // 2))


class _FixedEmbedding(nn.Module):
    def __init__(self, max_length, features):
        super().__init__()
        self.max_length = max_length
        self.embedding = nn.Embedding(max_length, features)


class _AdaLNParams(nn.Module):
    def __init__(self, style_dim, channels):
        super().__init__()
        self.fc = PlainLinear(style_dim, 2 * channels)


class _AttnOut(nn.Module):
    def __init__(self, mid, features):
        super().__init__()
        self.to_out = PlainLinear(mid, features)


class _Attention(nn.Module):
    def __init__(self, features, mid, style_dim=None):
        super().__init__()
        if style_dim is None:
            self.norm = nn.LayerNorm(features)
            self.norm_context = nn.LayerNorm(features)
        else:
            self.norm = _AdaLNParams(style_dim, features)
            self.norm_context = _AdaLNParams(style_dim, features)
        self.to_q = PlainLinear(features, mid, bias=False)
        self.to_kv = PlainLinear(features, 2 * mid, bias=False)
        self.attention = _AttnOut(mid, features)


class _Block(nn.Module):
    def __init__(self, features, mid, multiplier, style_dim=None):
        super().__init__()
        self.attention = _Attention(features, mid, style_dim)
        self.feed_forward = nn.Sequential(PlainLinear(features, features * multiplier), nn.Identity(),
                                          PlainLinear(features * multiplier, features))


@transient_state
class _Transformer(nn.Module):
    multispeaker = False

    def __init__(self, num_layers, channels, num_heads, head_features, multiplier, context_embedding_features,
                 context_features=None, embedding_max_length=512, **_unused):
        super().__init__()
        self.channels, self.heads, self.head_features = channels, num_heads, head_features
        self.features = channels + context_embedding_features
        self.emb_features = context_embedding_features
        mid = num_heads * head_features
        style = context_features if self.multispeaker else None
        self.blocks = nn.ModuleList([_Block(self.features, mid, multiplier, style) for _ in range(num_layers)])
        self.to_out = nn.Sequential(nn.Identity(), PlainConv1d(self.features, channels, 1))
        F_ = self.features
        self.to_mapping = nn.Sequential(PlainLinear(F_, F_), nn.Identity(), PlainLinear(F_, F_), nn.Identity())
        self.to_time = nn.Sequential(nn.Sequential(_LearnedPositionalEmbedding(channels), PlainLinear(channels + 1, F_)),
                                     nn.Identity())
        if self.multispeaker:
            self.to_features = nn.Sequential(PlainLinear(context_features, F_), nn.Identity())
        self.fixed_embedding = _FixedEmbedding(embedding_max_length, context_embedding_features)
        self._pk = None

    # -- packed weights --------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._pk = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._pk = None
        return super().load_state_dict(W.strip_module_prefix(state_dict), *a, **k)

    def _load_from_state_dict(self, *a, **k):  # also reached when a PARENT module's load_state_dict() recurses here
        self._pk = None                        # (models.load_checkpoint loads `diffusion` at the shell level)
        return super()._load_from_state_dict(*a, **k)

    def refresh(self):
        self._pk = None

    def __setattr__(self, name, value):
        if name == "_pk" and value is None:  # every invalidation of the Python-side pack (load_state_dict, .to(), refresh)
            object.__setattr__(self, "_engine_stale", True)   # ... makes the C++ plan's packed copy stale too
            object.__setattr__(self, "_pack_gen", getattr(self, "_pack_gen", 0) + 1)  # ... and recorded graphs
        super().__setattr__(name, value)

    def _prepare(self, device):
        d = lambda t: t.detach().float().contiguous().to(device)
        lin_t = lambda l: d(l.weight.detach().t())  # [in, out] for st2_style_fc
        pk = type("PackedDenoiser", (), {})()
        pk.device = device
        pk.time_w = d(self.to_time[0][0].weights)
        pk.time_lin, pk.time_b = lin_t(self.to_time[0][1]), d(self.to_time[0][1].bias)
        pk.map0, pk.map0_b = lin_t(self.to_mapping[0]), d(self.to_mapping[0].bias)
        pk.map2, pk.map2_b = lin_t(self.to_mapping[2]), d(self.to_mapping[2].bias)
        if self.multispeaker:
            pk.feat, pk.feat_b = lin_t(self.to_features[0]), d(self.to_features[0].bias)
            # every AdaLayerNorm fc of the net in one [style, J] matrix (they all consume `features`)
            fcs = [m for blk in self.blocks for m in (blk.attention.norm, blk.attention.norm_context)]
            pk.ada_wt = torch.cat([d(m.fc.weight).t() for m in fcs], dim=1).contiguous()
            pk.ada_b = torch.cat([d(m.fc.bias) for m in fcs]).contiguous()
        pk.blocks = []
        for blk in self.blocks:
            b = type("PackedBlock", (), {})()
            a = blk.attention
            if not self.multispeaker:
                b.n_w, b.n_b = d(a.norm.weight).reshape(1, -1), d(a.norm.bias).reshape(1, -1)
                b.nc_w, b.nc_b = d(a.norm_context.weight).reshape(1, -1), d(a.norm_context.bias).reshape(1, -1)
            b.q = W.pack_linear_auto(a.to_q.weight.detach().float()).to(device)
            b.kv = W.pack_linear_auto(a.to_kv.weight.detach().float()).to(device)
            b.o, b.o_b = W.pack_linear_auto(a.attention.to_out.weight.detach().float()).to(device), d(a.attention.to_out.bias)
            b.f1, b.f1_b = W.pack_linear_auto(blk.feed_forward[0].weight.detach().float()).to(device), d(blk.feed_forward[0].bias)
            b.f1_out = blk.feed_forward[0].weight.shape[0]
            b.f2, b.f2_b = W.pack_linear_auto(blk.feed_forward[2].weight.detach().float()).to(device), d(blk.feed_forward[2].bias)
            pk.blocks.append(b)
        pk.out_t = d(self.to_out[1].weight.detach().reshape(self.channels, self.features).t())
        pk.out_b = d(self.to_out[1].bias)
        pk.fixed = d(self.fixed_embedding.embedding.weight)
        self._pk = pk
        return pk

    # -- sessions: everything constant across the 2*(steps-1) net calls of one sampler run -------------
    def open_session(self, x, embedding=None, features=None, embedding_scale=1.0, embedding_mask_proba=0.0,
                     lengths=None):
        """`lengths` (host or device int tensor [B], optional): token count of every utterance of a right-padded batch.
        Attention then excludes the pad keys and the output mean runs over each utterance's own tokens, so every row
        equals that utterance's un-padded run (the reference runs one utterance at a time: no padding exists there)."""
        assert embedding is not None, "the denoiser is conditional: `embedding` is required (modules.py:410)"
        dev = x.device
        pk = self._pk if (self._pk is not None and self._pk.device == dev) else self._prepare(dev)
        B, N, E = embedding.shape
        assert N <= self.fixed_embedding.max_length, "Input sequence length must be <= max_length"
        assert E == self.emb_features
        s = type("DenoiserSession", (), {})()
        s.pk, s.B, s.N, s.scale = pk, B, N, float(embedding_scale)
        s.key_len = None
        if lengths is not None:
            assert lengths.numel() == B
            if lengths.dtype == torch.int32 and lengths.device == dev:
                s.key_len = lengths.contiguous()  # used as is: no host read (legal under hipGraph capture)
            else:
                assert int(lengths.min()) >= 1 and int(lengths.max()) <= N
                s.key_len = lengths.to(torch.int32).to(dev).contiguous()
        embedding = embedding.float()
        fixed = pk.fixed[:N].unsqueeze(0).expand(B, -1, -1)
        if embedding_mask_proba > 0.0:  # modules.py:412-416 (classifier-free-guidance dropout; off at inference)
            mask = torch.rand(B, 1, 1, device=dev) < embedding_mask_proba
            embedding = torch.where(mask, fixed, embedding)
        C = self.channels

        # Token-merged layout: storage is [F, B*N] -- channel-major over ALL tokens of the batch -- so every Linear is ONE
        # k=1 conv over B*N contiguous columns (25 full 128-column tiles at B=32, N=100 instead of 32 x one 78%-full
        # tile), while attention / norm statistics / the mapping add see the same memory as [B, F, N] through strides.
        # The multi-speaker net's AdaLayerNorm affine is per utterance: its q / kv convs take the [B, F, N] view of the same
        # storage, the other three Linears of a block (three quarters of the FLOPs) run merged like the single-speaker net's.
        s.merged = True

        def base(e):  # channel-major [x | embedding] buffer; rows < C are rewritten on every net call
            buf = self._alloc(s, self.features)
            ops.tokens_to_channels(e, buf[:, C:], B=B)
            return buf

        s.bases = [base(embedding.contiguous())]
        if s.scale != 1.0:
            s.bases.append(base(pk.fixed[:N]) if embedding_mask_proba <= 0.0 else base(fixed.contiguous()))
        s.feat_map, s.ada = None, None
        if self.multispeaker:
            assert features is not None, "context_features exists but no features provided"
            f = features.float().contiguous()
            s.feat_map = ops.style_fc(f, pk.feat, pk.feat_b, ops.ACT_GELU)
            s.ada = ops.style_fc(f, pk.ada_wt, pk.ada_b)
        return s

    def _mapping(self, s, c_noise):
        pk = s.pk
        four = ops.time_features(c_noise, pk.time_w, s.B)  # [t, sin(t w 2 pi), cos(t w 2 pi)], modules.py:666-671
        m = ops.style_fc(four, pk.time_lin, pk.time_b, ops.ACT_GELU)
        if s.feat_map is not None:
            m = ops.axpbypcz(m, 1.0, s.feat_map, 1.0)  # reduce(stack(items), 'sum'), modules.py:140
        m = ops.style_fc(m, pk.map0, pk.map0_b, ops.ACT_GELU)
        return ops.style_fc(m, pk.map2, pk.map2_b, ops.ACT_GELU)

    @staticmethod
    def _alloc(s, C):
        """[B, C, N] activation buffer; in the merged layout a strided view of a [C, B*N] block."""
        dev = s.pk.device
        if s.merged:
            return torch.empty((C, s.B, s.N), device=dev, dtype=torch.float32).permute(1, 0, 2)
        return torch.empty((s.B, C, s.N), device=dev, dtype=torch.float32)

    @staticmethod
    def _cv(s, t):
        """The view of an activation buffer the k=1 convs consume: [1, C, B*N] (merged) or [B, C, N]."""
        if not s.merged:
            return t
        C = t.shape[1]
        return t.permute(1, 0, 2).reshape(1, C, s.B * s.N)  # a view: the storage is [C, B, N] contiguous

    def _run(self, s, base, x, m):
        pk = s.pk
        B, N, Fz, C = s.B, s.N, self.features, self.channels
        mid = self.heads * self.head_features
        A, V = (lambda c: self._alloc(s, c)), (lambda t: self._cv(s, t))
        ops.broadcast_cols(x.reshape(B, C), base[:, :C])
        X = ops.add_chanvec(base, m, out=A(Fz))
        nblk = len(pk.blocks)
        for i, b in enumerate(pk.blocks):
            st = ops.colnorm_stats(X)
            # The multispeaker net's AdaLayerNorm affine is per utterance.  With the token-merged storage its q / kv convs
            # still run as ONE GEMM over the B*N columns: st2_act_split takes the affine row of a column from its
            # utterance (gb_seg = N).  Needs the xs pair (>= XS_MIN_L columns); smaller calls keep the [B, F, N] view.
            # ... and a net wide enough that ops.conv1d does not route its k = 1 convs to the fused kernel (no gb_seg there)
            seg = (self.multispeaker and s.merged and B > 1 and B * N >= ops.XS_MIN_L
                   and not ops.prefer_fused(ops.PRO_COLNORM, Fz, 1))
            stv = st if (self.multispeaker and not seg) else st.view(1, B * N, 2)
            qkv = A(3 * mid)
            if self.multispeaker:
                o = 4 * Fz * i
                g1, b1 = s.ada[:, o:o + Fz], s.ada[:, o + Fz:o + 2 * Fz]
                g2, b2 = s.ada[:, o + 2 * Fz:o + 3 * Fz], s.ada[:, o + 3 * Fz:o + 4 * Fz]
                kw1 = dict(gamma=g1, beta=b1, gamma_plus_one=True, gb_seg=N if seg else 0)
                kw2 = dict(gamma=g2, beta=b2, gamma_plus_one=True, gb_seg=N if seg else 0)
            else:
                kw1 = dict(gamma=b.n_w, beta=b.n_b)
                kw2 = dict(gamma=b.nc_w, beta=b.nc_b)
            Vq = (lambda t: t) if (self.multispeaker and not seg) else V
            ops.conv1d(Vq(X), b.q, mid, 1, pro=ops.PRO_COLNORM, stats=stv, out=Vq(qkv)[:, :mid], **kw1)
            ops.conv1d(Vq(X), b.kv, 2 * mid, 1, pro=ops.PRO_COLNORM, stats=stv, out=Vq(qkv)[:, mid:], **kw2)
            att = ops.attention(qkv[:, :mid], qkv[:, mid:2 * mid], qkv[:, 2 * mid:], self.heads,
                                self.head_features ** -0.5, out=A(mid), key_len=s.key_len)
            X1, hmid, X2 = A(Fz), A(b.f1_out), A(Fz)
            ops.conv1d(V(att), b.o, Fz, 1, bias=b.o_b, res=V(X), out=V(X1))
            ops.conv1d(V(X1), b.f1, b.f1_out, 1, bias=b.f1_b, act=ops.ACT_GELU, out=V(hmid))
            ops.conv1d(V(hmid), b.f2, Fz, 1, bias=b.f2_b, res=V(X1), out=V(X2))
            X = ops.add_chanvec(X2, m, out=A(Fz)) if i + 1 < nblk else X2
        mean = ops.mean_tokens(X, lengths=s.key_len)  # mean over the utterance's tokens, modules.py:155,397
        return ops.style_fc(mean, pk.out_t, pk.out_b).reshape(B, 1, C)

    def run_session(self, s, x, c_noise):
        m = self._mapping(s, c_noise)
        out = self._run(s, s.bases[0], x, m)
        if s.scale != 1.0:  # classifier-free guidance, modules.py:418-423
            out_masked = self._run(s, s.bases[1], x, m)
            return ops.axpbypcz(out_masked, 1.0 - s.scale, out, s.scale)
        return out

    @torch.no_grad()
    def forward(self, x, time, embedding_mask_proba=0.0, embedding=None, features=None, embedding_scale=1.0,
                lengths=None):
        """Reference call signature (modules.py:402-407).  `time` must be batch-uniform (it is c_noise of a
        scalar sigma at inference)."""
        tv = time.reshape(-1)
        if tv.numel() > 1 and not bool((tv == tv[0]).all()):
            raise NotImplementedError("per-item time embeddings are a training-time feature")
        s = self.open_session(x, embedding=embedding, features=features, embedding_scale=embedding_scale,
                              embedding_mask_proba=embedding_mask_proba, lengths=lengths)
        return self.run_session(s, x.float().contiguous(), float(tv[0]))


class Transformer1d(_Transformer):
    """Single-speaker denoiser (modules.py:283-427): nn.LayerNorm before q / kv."""
    multispeaker = False


class StyleTransformer1d(_Transformer):
    """Multi-speaker denoiser (modules.py:40-185): AdaLayerNorm conditioned on `features`."""
    multispeaker = True


class AudioDiffusionConditional(nn.Module):
    """Shell that exposes the denoiser under both `diffusion.net.*` and `unet.*` state_dict prefixes
    (Modules/diffusion/diffusion.py + models.py:653-669)."""

    def __init__(self, transformer, sigma_data, embedding_mask_proba=0.1):
        super().__init__()
        self.diffusion = KDiffusion(net=transformer, sigma_data=sigma_data)
        self.unet = transformer
        self.embedding_mask_proba = embedding_mask_proba
