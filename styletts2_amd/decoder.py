"""Acoustic decoder + vocoder engine: drop-in for the reference `Decoder`
(Modules/istftnet.py:467-528 and Modules/hifigan.py:416-475, selected by `decoder.type` exactly as
models.py:617-633 does).

    decoder(asr[B,512,T], F0_curve[B,2T], N[B,2T], s[B,128]) -> wave[B,1,600*T]

The module keeps the reference's state_dict layout (layers.py) and, on first use after a load,
folds weight-norm and packs every conv into the pre-split f16 layout of `st2_conv1d_xs` / `st2_conv1d_f16s`
(`st2_conv1d`'s K-major fp32 layout under the tests' `_hooks.override(conv_precision="f32")`).  forward() is a straight-line plan of HIP
kernel launches on torch's current stream: no host synchronisation, no per-forward weight-norm, one batched
style-FC GEMM for all AdaIN layers; every AdaIN + Snake / LeakyReLU is one HBM-bound `st2_act_split` pass that
hands the following MFMA conv pre-split operands, and the InstanceNorm statistics it needs come out of the
producing conv's epilogue (`want_stats`), so no tensor is read just to be reduced.
"""
import math

import torch
import torch.nn as nn

from .layers import transient_state
from . import ops
from . import weights as W
from .layers import (AdaINParams, AdaINResBlock1Params, AdainResBlk1dParams, PlainConv1d, PlainLinear, WNConv1d,
                     WNConvTranspose1d)

LRELU_SLOPE = 0.1  # Modules/istftnet.py:13


class StyleBank:
    """All AdaIN `fc` layers of a module evaluated by ONE st2_style_fc launch per forward
    (they all consume the same style vector: Modules/istftnet.py:19-24, 58/106 layers per decoder)."""

    def __init__(self):
        self._mods, self._off = [], {}
        self.J = 0

    def add(self, adain: AdaINParams):
        self._off[id(adain)] = (self.J, adain.channels)
        self._mods.append(adain)
        self.J += 2 * adain.channels

    def pack(self, device):
        self.wt = torch.cat([m.fc.weight.detach().t() for m in self._mods], dim=1).contiguous().float().to(device)
        self.bias = torch.cat([m.fc.bias.detach() for m in self._mods]).contiguous().float().to(device)

    def run(self, s):
        return ops.style_fc(s.contiguous(), self.wt, self.bias)

    def gb(self, h, adain):
        off, c = self._off[id(adain)]
        return h[:, off:off + c], h[:, off + c:off + 2 * c]


def _dev(t, device):
    return t.detach().float().contiguous().to(device)


class _PackedConv:
    def __init__(self, conv: WNConv1d, device):
        self.wt = W.pack_conv_auto(conv.folded().float()).to(device)
        self.bias = _dev(conv.bias, device) if conv.bias is not None else None
        self.c_out, self.ks = conv.c_out, conv.ks


class _PackedResBlock1:
    def __init__(self, p: AdaINResBlock1Params, device):
        self.p = p
        self.c1 = [_PackedConv(c, device) for c in p.convs1]
        self.c2 = [_PackedConv(c, device) for c in p.convs2]
        self.a1 = [_dev(a.reshape(-1), device) for a in p.alpha1]
        self.a2 = [_dev(a.reshape(-1), device) for a in p.alpha2]


def run_resblock1(pk: _PackedResBlock1, bank: StyleBank, h, x, x_stats=None, mrf_acc=None, mrf_last=False,
                  n_mrf=3, out=None):
    """AdaINResBlock1.forward (Modules/istftnet.py:66-75).  `mrf_acc`/`mrf_last` fold the multi-receptive-field
    sum `xs += resblock(x)` and the final `/ num_kernels` (istftnet.py:369-375) into the last conv's epilogue."""
    p = pk.p
    C, ks = p.channels, p.ks
    nsteps = len(p.dilation)
    st = x_stats if x_stats is not None else ops.instnorm_stats(x)
    for i, d in enumerate(p.dilation):
        g1, b1 = bank.gb(h, p.adain1[i])
        # the InstanceNorm statistics each AdaIN needs come out of the producing conv's epilogue (want_stats)
        xt, st2 = ops.conv1d(x, pk.c1[i].wt, C, ks, dil=d, pad_left=(ks * d - d) // 2, bias=pk.c1[i].bias,
                             pro=ops.PRO_ADAIN_SNAKE, stats=st, gamma=g1, beta=b1, alpha=pk.a1[i], want_stats=True)
        g2, b2 = bank.gb(h, p.adain2[i])
        last = i == nsteps - 1
        kw = dict(dil=1, pad_left=(ks - 1) // 2, bias=pk.c2[i].bias, pro=ops.PRO_ADAIN_SNAKE, stats=st2, gamma=g2,
                  beta=b2, alpha=pk.a2[i], res=x)
        if last:
            x = ops.conv1d(xt, pk.c2[i].wt, C, ks, res2=mrf_acc, div=float(n_mrf) if mrf_last else 1.0, out=out, **kw)
        else:
            x, st = ops.conv1d(xt, pk.c2[i].wt, C, ks, want_stats=True, **kw)
    return x


class _PackedAdainResBlk:
    def __init__(self, p: AdainResBlk1dParams, device):
        self.p = p
        self.conv1 = _PackedConv(p.conv1, device)
        self.conv2 = _PackedConv(p.conv2, device)
        self.sc = _PackedConv(p.conv1x1, device) if p.learned_sc else None
        if p.upsample:
            self.pool_w = _dev(p.pool.folded().reshape(p.dim_in, 3), device)
            self.pool_b = _dev(p.pool.bias, device)


def run_adain_resblk(pk: _PackedAdainResBlk, bank: StyleBank, h, x, out=None):
    """AdainResBlk1d.forward (Modules/istftnet.py:435-454): (residual(x, s) + shortcut(x)) / sqrt(2).
    The 1x1 shortcut commutes with nearest x2 up-sampling, so it runs at the low rate and the final conv's
    epilogue reads it with `l >> 1`."""
    p = pk.p
    st1 = ops.instnorm_stats(x)
    g1, b1 = bank.gb(h, p.norm1)
    if p.upsample:
        u = ops.adain_leaky_pool(x, st1, g1, b1, 0.2, pk.pool_w, pk.pool_b)
        t1, st2 = ops.conv1d(u, pk.conv1.wt, p.dim_out, 3, pad_left=1, bias=pk.conv1.bias, want_stats=True)
    else:
        t1, st2 = ops.conv1d(x, pk.conv1.wt, p.dim_out, 3, pad_left=1, bias=pk.conv1.bias, pro=ops.PRO_ADAIN_LEAKY,
                             slope=0.2, stats=st1, gamma=g1, beta=b1, want_stats=True)
    g2, b2 = bank.gb(h, p.norm2)
    sc = ops.conv1d(x, pk.sc.wt, p.dim_out, 1) if pk.sc is not None else x
    return ops.conv1d(t1, pk.conv2.wt, p.dim_out, 3, pad_left=1, bias=pk.conv2.bias, pro=ops.PRO_ADAIN_LEAKY,
                      slope=0.2, stats=st2, gamma=g2, beta=b2, res=sc, res_shift=1 if p.upsample else 0,
                      div=math.sqrt(2), out=out)


class _SourceModule(nn.Module):
    """SourceModuleHnNSF state (Modules/istftnet.py:250-297): only `l_linear` has parameters."""

    def __init__(self, harmonics):
        super().__init__()
        self.l_linear = PlainLinear(harmonics, 1)


class Generator(nn.Module):
    """iSTFTNet generator (Modules/istftnet.py:302-380) or HiFi-GAN generator (Modules/hifigan.py:272-347)."""

    HARMONICS = 9

    def __init__(self, kind, style_dim, resblock_kernel_sizes, upsample_rates, upsample_initial_channel,
                 resblock_dilation_sizes, upsample_kernel_sizes, gen_istft_n_fft=None, gen_istft_hop_size=None):
        super().__init__()
        assert kind in ("istftnet", "hifigan")
        self.kind = kind
        self.rates, self.up_ks = list(upsample_rates), list(upsample_kernel_sizes)
        self.num_kernels, self.num_upsamples = len(resblock_kernel_sizes), len(upsample_rates)
        self.n_fft, self.hop = gen_istft_n_fft, gen_istft_hop_size
        c0 = upsample_initial_channel
        self.channels = [c0 // (2 ** (i + 1)) for i in range(self.num_upsamples)]
        self.m_source = _SourceModule(self.HARMONICS)
        src_ch = (self.n_fft + 2) if kind == "istftnet" else 1
        self.noise_convs, self.noise_res, self.ups, self.resblocks = (nn.ModuleList(), nn.ModuleList(),
                                                                      nn.ModuleList(), nn.ModuleList())
        for i, (u, k) in enumerate(zip(self.rates, self.up_ks)):
            c_cur = self.channels[i]
            self.ups.append(WNConvTranspose1d(c0 // (2 ** i), c_cur, k, c_cur))
            if i + 1 < self.num_upsamples:
                stride_f0 = int(math.prod(self.rates[i + 1:]))
                self.noise_convs.append(PlainConv1d(src_ch, c_cur, stride_f0 * 2))
                self.noise_res.append(AdaINResBlock1Params(c_cur, 7, (1, 3, 5), style_dim))
            else:
                self.noise_convs.append(PlainConv1d(src_ch, c_cur, 1))
                self.noise_res.append(AdaINResBlock1Params(c_cur, 11, (1, 3, 5), style_dim))
        if kind == "hifigan":
            self.alphas = nn.ParameterList([nn.Parameter(torch.ones(1, c0, 1))] +
                                           [nn.Parameter(torch.ones(1, c, 1)) for c in self.channels])
        for i in range(self.num_upsamples):
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(AdaINResBlock1Params(self.channels[i], k, tuple(d), style_dim))
        self.conv_post = WNConv1d(self.channels[-1], (self.n_fft + 2) if kind == "istftnet" else 1, 7)
        self.upsample_scale = int(math.prod(self.rates)) * (self.hop if kind == "istftnet" else 1)

    # -- plan helpers ----------------------------------------------------------------------
    def register(self, bank: StyleBank):
        for rb in list(self.noise_res) + list(self.resblocks):
            for a in list(rb.adain1) + list(rb.adain2):
                bank.add(a)

    def pack(self, device):
        pk = type("PackedGenerator", (), {})()
        pk.lin_w = _dev(self.m_source.l_linear.weight.reshape(-1), device)
        pk.lin_b = _dev(self.m_source.l_linear.bias.reshape(-1), device)
        # noise_convs (istftnet.py:332-339): kernel = 2*stride -> polyphase k=2 conv on the matrix pipe; last one is k=1
        pk.noise_wt, pk.noise_stride = [], []
        for i, c in enumerate(self.noise_convs):
            stride_f0 = int(math.prod(self.rates[i + 1:])) if i + 1 < self.num_upsamples else 1
            w = c.weight.detach().float()
            if stride_f0 > 1:
                w = W.polyphase_strided_conv(w, stride_f0)
            pk.noise_wt.append(W.pack_conv_auto(w).to(device))
            pk.noise_stride.append(stride_f0)
        pk.noise_b = [_dev(c.bias, device) for c in self.noise_convs]
        pk.noise_res = [_PackedResBlock1(r, device) for r in self.noise_res]
        pk.resblocks = [_PackedResBlock1(r, device) for r in self.resblocks]
        pk.ups_wt = [W.pack_conv_auto(W.polyphase_convt(u.folded().float(), r)).to(device)
                     for u, r in zip(self.ups, self.rates)]
        pk.ups_b = [_dev(u.bias, device) for u in self.ups]
        pk.post = _PackedConv(self.conv_post, device)
        if self.kind == "hifigan":
            pk.alphas = [_dev(a.reshape(-1), device) for a in self.alphas]
        return pk

    def _mrf(self, pk, bank, h, x, st, i):
        """Multi-receptive-field fusion: num_kernels AdaINResBlock1 chains on the same input, summed and divided.  The
        chains only meet in the running sum, which the LAST conv of each chain adds in its epilogue; the summation order
        is ((r0 + r1) + r2) / n.  (Issuing the chains on separate streams was measured in round 1 and is neutral: a
        12 000-workgroup conv leaves no CU slots to co-run in -- DESIGN.md section 3.)"""
        n = self.num_kernels
        blocks = pk.resblocks[i * n:(i + 1) * n]
        acc = None
        for j in range(n):
            acc = run_resblock1(blocks[j], bank, h, x, x_stats=st, mrf_acc=acc, mrf_last=(j == n - 1), n_mrf=n)
        return acc

    def run(self, pk, bank, h, x, f0_curve, noise=None, har=None, taps=None):
        B = x.shape[0]
        L = f0_curve.shape[1] * self.upsample_scale
        ist = self.kind == "istftnet"
        if har is None:
            if noise is None:
                noise = torch.randn(B, L, self.HARMONICS, device=x.device, dtype=torch.float32)
            har_source = ops.har_source(f0_curve.contiguous(), self.upsample_scale, noise, pk.lin_w, pk.lin_b,
                                        sine_amp=0.1, noise_std=0.003, voiced_threshold=10.0, sample_rate=24000.0)
            har = ops.stft_mag_phase(har_source, self.n_fft, self.hop) if ist else har_source.unsqueeze(1)
            if taps is not None:
                taps["har_source"] = har_source
        if taps is not None:
            taps["har"] = har
        for i in range(self.num_upsamples):
            u, k, C = self.rates[i], self.up_ks[i], self.channels[i]
            last = i == self.num_upsamples - 1
            # harmonic-source branch (istftnet.py:361-362 / hifigan.py:330-331)
            stride_f0 = pk.noise_stride[i]
            if stride_f0 > 1:
                pad_f0 = (stride_f0 + 1) // 2
                L_ns = (har.shape[2] + 2 * pad_f0 - 2 * stride_f0) // stride_f0 + 1
                harp = ops.phase_split(har, stride_f0, pad_f0, L_ns + 1)
                xs = ops.conv1d(harp, pk.noise_wt[i], C, 2, pad_left=0, L_out=L_ns, bias=pk.noise_b[i])
            else:
                xs = ops.conv1d(har, pk.noise_wt[i], C, 1, bias=pk.noise_b[i])
            xs = run_resblock1(pk.noise_res[i], bank, h, xs)
            # up-sampling ConvTranspose1d as polyphase GEMM + interleave (istftnet.py:360,364-368)
            L_in = x.shape[2]
            if ist:
                pad, L_raw = (k - u) // 2, (L_in - 1) * u - 2 * ((k - u) // 2) + k
                pro = dict(pro=ops.PRO_LEAKY, slope=LRELU_SLOPE)
            else:
                pad, L_raw = u // 2 + u % 2, (L_in - 1) * u - 2 * (u // 2 + u % 2) + k + u % 2
                pro = dict(pro=ops.PRO_SNAKE, alpha=pk.alphas[i])
            Y = ops.conv1d(x, pk.ups_wt[i], u * C, 2, pad_left=1, L_out=L_in + 1, **pro)
            x, st = ops.convt_interleave(Y, C, u, pad, L_raw, bias=pk.ups_b[i], add=xs, reflect_left=(ist and last),
                                         want_stats=True)  # statistics for the MRF's first AdaINs ride along
            # multi-receptive-field fusion (istftnet.py:369-375)
            x = self._mrf(pk, bank, h, x, st, i)
            if taps is not None:
                taps["stage%d" % i] = x
        if ist:
            nb = self.n_fft // 2 + 1
            sp = ops.conv1d(x, pk.post.wt, self.n_fft + 2, 7, pad_left=3, bias=pk.post.bias, pro=ops.PRO_LEAKY,
                            slope=0.01, act=ops.ACT_EXP_SIN, act_split=nb)  # F.leaky_relu default slope, :376
            if taps is not None:
                taps["spec_phase"] = sp
            return ops.istft(sp, self.n_fft, self.hop)
        return ops.conv1d(x, pk.post.wt, 1, 7, pad_left=3, bias=pk.post.bias, pro=ops.PRO_SNAKE,
                          alpha=pk.alphas[self.num_upsamples], act=ops.ACT_TANH)


@transient_state
class Decoder(nn.Module):
    """Drop-in for the reference Decoder (both vocoder variants)."""

    def __init__(self, dim_in=512, F0_channel=512, style_dim=64, dim_out=80, resblock_kernel_sizes=(3, 7, 11),
                 upsample_rates=(10, 6), upsample_initial_channel=512,
                 resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)), upsample_kernel_sizes=(20, 12),
                 gen_istft_n_fft=20, gen_istft_hop_size=5, kind="istftnet"):
        super().__init__()
        self.kind, self.dim_in = kind, dim_in
        self.encode = AdainResBlk1dParams(dim_in + 2, 1024, style_dim)
        self.decode = nn.ModuleList([AdainResBlk1dParams(1024 + 2 + 64, 1024, style_dim) for _ in range(3)] +
                                    [AdainResBlk1dParams(1024 + 2 + 64, 512, style_dim, upsample=True)])
        self.F0_conv = WNConv1d(1, 1, 3)
        self.N_conv = WNConv1d(1, 1, 3)
        self.asr_res = nn.Sequential(WNConv1d(dim_in, 64, 1))
        if kind == "istftnet":
            self.generator = Generator(kind, style_dim, resblock_kernel_sizes, upsample_rates,
                                       upsample_initial_channel, resblock_dilation_sizes, upsample_kernel_sizes,
                                       gen_istft_n_fft, gen_istft_hop_size)
        else:
            self.generator = Generator(kind, style_dim, resblock_kernel_sizes, upsample_rates,
                                       upsample_initial_channel, resblock_dilation_sizes, upsample_kernel_sizes)
        self._pk = None

    def __setattr__(self, name, value):
        if name == "_pk" and value is None:  # every invalidation of the Python-side pack drops the C++ plan's copy too
            old = self.__dict__.get("_eng")
            if old is not None:
                from . import engine
                engine.replaced(old, "decoder")  # a calibrated engine does not vanish silently (load_state_dict / .to())
            object.__setattr__(self, "_eng", None)
        super().__setattr__(name, value)

    # -- packed-weight cache ---------------------------------------------------------------
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
        """Call after mutating parameters in place."""
        self._pk = None

    def _prepare(self, device):
        pk = type("PackedDecoder", (), {})()
        pk.device = device
        bank = StyleBank()
        for blk in [self.encode] + list(self.decode):
            bank.add(blk.norm1)
            bank.add(blk.norm2)
        self.generator.register(bank)
        bank.pack(device)
        pk.bank = bank
        pk.encode = _PackedAdainResBlk(self.encode, device)
        pk.decode = [_PackedAdainResBlk(b, device) for b in self.decode]
        pk.f0_w, pk.f0_b = _dev(self.F0_conv.folded(), device), _dev(self.F0_conv.bias, device)
        pk.n_w, pk.n_b = _dev(self.N_conv.folded(), device), _dev(self.N_conv.bias, device)
        pk.asr_res = _PackedConv(self.asr_res[0], device)
        pk.gen = self.generator.pack(device)
        self._pk = pk
        return pk

    @torch.no_grad()
    def forward(self, asr, F0_curve, N, s, noise=None, har=None, taps=None):
        """`noise` [B, 600*T, 9] replaces the reference's in-forward `torch.randn_like` draw
        (Modules/istftnet.py:242) so parity runs can replay the oracle's tensor; `har` injects the
        harmonic STFT features (tap-point protocol); `taps` (dict) collects
        intermediates."""
        if self.training:
            raise RuntimeError("the MI355X engine is inference-only; call .eval() (reference: istftnet.py:500-508 "
                               "is the training-only F0/N smoothing)")
        dev = asr.device
        from . import engine
        if asr.is_cuda and engine.plan_mode() == "engine" and W.conv_precision() == "f16s":
            # ONE C-ABI call: the launch plan below exists in C++ (csrc/st2_engine.hip, st2_decoder_forward)
            eng = self._eng
            if eng is None or not engine.same_device(eng, dev):
                engine.replaced(eng, "decoder")
                eng = self._eng = engine.build_decoder_engine(self, dev)
            return eng.decoder_forward(asr, F0_curve, N, s, noise=noise, har=har, taps=taps)
        pk = self._pk if (self._pk is not None and self._pk.device == dev) else self._prepare(dev)
        bank = pk.bank
        asr = asr.float().contiguous()
        B, Cin, T = asr.shape
        h = bank.run(s.float())
        F0_curve = F0_curve.float().contiguous()
        N = N.float().contiguous()
        # [x(1024) | asr_res(64) | F0 | N] lives in one buffer; producers write their channel slices in place
        cat = torch.empty((B, 1024 + 64 + 2, T), device=dev, dtype=torch.float32)
        cat0 = torch.empty((B, Cin + 2, T), device=dev, dtype=torch.float32)
        ops.copy_ncl(asr, cat0[:, :Cin])
        ops.conv1d_direct(F0_curve.unsqueeze(1), pk.f0_w, pk.f0_b, 2, 1, out=cat0[:, Cin:Cin + 1])
        ops.conv1d_direct(N.unsqueeze(1), pk.n_w, pk.n_b, 2, 1, out=cat0[:, Cin + 1:Cin + 2])
        ops.copy_ncl(cat0[:, Cin:Cin + 2], cat[:, 1088:1090])
        ops.conv1d(asr, pk.asr_res.wt, 64, 1, bias=pk.asr_res.bias, out=cat[:, 1024:1088])
        run_adain_resblk(pk.encode, bank, h, cat0, out=cat[:, :1024])
        if taps is not None:
            taps["encode"] = cat[:, :1024].clone()
        x = None
        for i, blk in enumerate(pk.decode):
            if blk.p.upsample:
                x = run_adain_resblk(blk, bank, h, cat)
            else:
                run_adain_resblk(blk, bank, h, cat, out=cat[:, :1024])
        if taps is not None:
            taps["front"] = x
        return self.generator.run(pk.gen, bank, h, x, F0_curve, noise=noise, har=har, taps=taps)
