"""ORACLE -- test infrastructure only.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product path (styletts2_amd/) never does.

A CPU restatement, in plain functional PyTorch fp32 ops, of the reference's text->waveform hot path.
Every function cites the reference file:line it follows.  It consumes a *reference-layout*
state_dict (the same one the engine loads), so oracle and engine always see identical weights.

Pinning: the reference ships no tests or golden vectors, so this restatement is
pinned against outputs of the reference's own modules executed in the build container
(oracle/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).  The arithmetic
under it is third-party and unpinned by the reference: torch (requirements.txt:4, here 2.10.0) and
scipy.signal.get_window (Modules/istftnet.py:89).
"""
import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # Modules/istftnet.py:13


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def wn(sd, prefix):
    """old-style weight_norm: w = g * v / ||v|| over all dims but 0 (torch.nn.utils.weight_norm, dim=0)."""
    g, v = sd[prefix + ".weight_g"], sd[prefix + ".weight_v"]
    return v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1))))


def sub(sd, prefix):
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


# ------------------------------------------------------------------------------------------------
# AdaIN blocks
# ------------------------------------------------------------------------------------------------
def adain1d(sd, prefix, x, s):
    """AdaIN1d.forward, Modules/istftnet.py:21-25."""
    h = F.linear(s, sd[prefix + ".fc.weight"], sd[prefix + ".fc.bias"])
    gamma, beta = torch.chunk(h.unsqueeze(-1), 2, dim=1)
    return (1 + gamma) * F.instance_norm(x, eps=1e-5) + beta


def snake(x, alpha):
    """Modules/istftnet.py:69."""
    return x + (1 / alpha) * (torch.sin(alpha * x) ** 2)


def adain_resblock1(sd, prefix, x, s, ks, dilation=(1, 3, 5)):
    """AdaINResBlock1.forward, Modules/istftnet.py:66-75."""
    for i, d in enumerate(dilation):
        xt = adain1d(sd, "%s.adain1.%d" % (prefix, i), x, s)
        xt = snake(xt, sd["%s.alpha1.%d" % (prefix, i)])
        xt = F.conv1d(xt, wn(sd, "%s.convs1.%d" % (prefix, i)), sd["%s.convs1.%d.bias" % (prefix, i)],
                      dilation=d, padding=(ks * d - d) // 2)
        xt = adain1d(sd, "%s.adain2.%d" % (prefix, i), xt, s)
        xt = snake(xt, sd["%s.alpha2.%d" % (prefix, i)])
        xt = F.conv1d(xt, wn(sd, "%s.convs2.%d" % (prefix, i)), sd["%s.convs2.%d.bias" % (prefix, i)],
                      padding=(ks - 1) // 2)
        x = xt + x
    return x


def adain_resblk1d(sd, prefix, x, s):
    """AdainResBlk1d.forward, Modules/istftnet.py:435-454 (same code models.py:372-416)."""
    upsample = (prefix + ".pool.weight_g") in sd
    # residual branch
    r = adain1d(sd, prefix + ".norm1", x, s)
    r = F.leaky_relu(r, 0.2)
    if upsample:
        C = x.shape[1]
        r = F.conv_transpose1d(r, wn(sd, prefix + ".pool"), sd[prefix + ".pool.bias"], stride=2, padding=1,
                               output_padding=1, groups=C)
    r = F.conv1d(r, wn(sd, prefix + ".conv1"), sd[prefix + ".conv1.bias"], padding=1)
    r = adain1d(sd, prefix + ".norm2", r, s)
    r = F.leaky_relu(r, 0.2)
    r = F.conv1d(r, wn(sd, prefix + ".conv2"), sd[prefix + ".conv2.bias"], padding=1)
    # shortcut branch
    sc = x
    if upsample:
        sc = F.interpolate(sc, scale_factor=2, mode="nearest")
    if (prefix + ".conv1x1.weight_g") in sd:
        sc = F.conv1d(sc, wn(sd, prefix + ".conv1x1"))
    return (r + sc) / math.sqrt(2)


# ------------------------------------------------------------------------------------------------
# harmonic source, STFT, iSTFT
# ------------------------------------------------------------------------------------------------
def sine_source(sd, prefix, f0_curve, upsample_scale, noise, harmonics=9, sine_amp=0.1, noise_std=0.003,
                voiced_threshold=10.0, sample_rate=24000):
    """Generator.forward head + SourceModuleHnNSF + SineGen, Modules/istftnet.py:352-354,283-297,141-247.
    `noise` [B, L, 9] is the randn_like draw of istftnet.py:242 made explicit.  `rand_ini` (:155-158) cannot
    reach the output and is omitted."""
    f0 = F.interpolate(f0_curve[:, None], scale_factor=float(upsample_scale), mode="nearest").transpose(1, 2)
    fn = f0 * torch.arange(1, harmonics + 1, dtype=torch.float32).view(1, 1, -1)
    rad = (fn / sample_rate) % 1
    rad = F.interpolate(rad.transpose(1, 2), scale_factor=1 / upsample_scale, mode="linear").transpose(1, 2)
    phase = torch.cumsum(rad, dim=1) * 2 * math.pi
    phase = F.interpolate(phase.transpose(1, 2) * upsample_scale, scale_factor=float(upsample_scale),
                          mode="linear").transpose(1, 2)
    sines = torch.sin(phase) * sine_amp
    uv = (f0 > voiced_threshold).float()
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3
    sine_waves = sines * uv + noise_amp * noise
    merged = torch.tanh(F.linear(sine_waves, sd[prefix + ".l_linear.weight"], sd[prefix + ".l_linear.bias"]))
    return merged.transpose(1, 2).squeeze(1)  # [B, L]


def hann(n):
    """scipy.signal.get_window('hann', n, fftbins=True) (Modules/istftnet.py:89): periodic Hann."""
    return (0.5 - 0.5 * torch.cos(2 * math.pi * torch.arange(n, dtype=torch.float64) / n)).float()


def stft_mag_phase(x, n_fft, hop):
    """TorchSTFT.transform, Modules/istftnet.py:91-97."""
    X = torch.stft(x, n_fft, hop, n_fft, window=hann(n_fft), return_complex=True)
    return torch.abs(X), torch.angle(X)


def istft(mag, phase, n_fft, hop):
    """TorchSTFT.inverse, Modules/istftnet.py:99-104."""
    y = torch.istft(mag * torch.exp(phase * 1j), n_fft, hop, n_fft, window=hann(n_fft))
    return y.unsqueeze(-2)


# ------------------------------------------------------------------------------------------------
# generators and decoder
# ------------------------------------------------------------------------------------------------
def generator_istftnet(sd, cfg, x, s, f0_curve, noise=None, har=None, taps=None):
    """Generator.forward, Modules/istftnet.py:350-380."""
    rates, up_ks = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    rks, rds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    n_fft, hop = cfg["gen_istft_n_fft"], cfg["gen_istft_hop_size"]
    nk, nu = len(rks), len(rates)
    if har is None:
        har_source = sine_source(sd, "m_source", f0_curve, int(math.prod(rates)) * hop, noise)
        spec, ph = stft_mag_phase(har_source, n_fft, hop)
        har = torch.cat([spec, ph], dim=1)
        if taps is not None:
            taps["har_source"] = har_source
    if taps is not None:
        taps["har"] = har
    for i in range(nu):
        x = F.leaky_relu(x, LRELU_SLOPE)
        if i + 1 < nu:
            st = int(math.prod(rates[i + 1:]))
            x_source = F.conv1d(har, sd["noise_convs.%d.weight" % i], sd["noise_convs.%d.bias" % i], stride=st,
                                padding=(st + 1) // 2)
            x_source = adain_resblock1(sd, "noise_res.%d" % i, x_source, s, 7)
        else:
            x_source = F.conv1d(har, sd["noise_convs.%d.weight" % i], sd["noise_convs.%d.bias" % i])
            x_source = adain_resblock1(sd, "noise_res.%d" % i, x_source, s, 11)
        x = F.conv_transpose1d(x, wn(sd, "ups.%d" % i), sd["ups.%d.bias" % i], stride=rates[i],
                               padding=(up_ks[i] - rates[i]) // 2)
        if i == nu - 1:
            x = F.pad(x, (1, 0), mode="reflect")
        x = x + x_source
        xs = None
        for j in range(nk):
            r = adain_resblock1(sd, "resblocks.%d" % (i * nk + j), x, s, rks[j], tuple(rds[j]))
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps["stage%d" % i] = x
    x = F.leaky_relu(x)  # default slope 0.01, istftnet.py:376
    x = F.conv1d(x, wn(sd, "conv_post"), sd["conv_post.bias"], padding=3)
    nb = n_fft // 2 + 1
    spec = torch.exp(x[:, :nb])
    phase = torch.sin(x[:, nb:])
    if taps is not None:
        taps["spec_phase"] = torch.cat([spec, phase], dim=1)
    return istft(spec, phase, n_fft, hop)


def generator_hifigan(sd, cfg, x, s, f0_curve, noise=None, har=None, taps=None):
    """Generator.forward, Modules/hifigan.py:321-347."""
    rates, up_ks = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    rks, rds = cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]
    nk, nu = len(rks), len(rates)
    if har is None:
        har = sine_source(sd, "m_source", f0_curve, int(math.prod(rates)), noise).unsqueeze(1)
        if taps is not None:
            taps["har_source"] = har.squeeze(1)
    if taps is not None:
        taps["har"] = har
    for i in range(nu):
        x = snake(x, sd["alphas.%d" % i])
        u = rates[i]
        if i + 1 < nu:
            st = int(math.prod(rates[i + 1:]))
            x_source = F.conv1d(har, sd["noise_convs.%d.weight" % i], sd["noise_convs.%d.bias" % i], stride=st,
                                padding=(st + 1) // 2)
            x_source = adain_resblock1(sd, "noise_res.%d" % i, x_source, s, 7)
        else:
            x_source = F.conv1d(har, sd["noise_convs.%d.weight" % i], sd["noise_convs.%d.bias" % i])
            x_source = adain_resblock1(sd, "noise_res.%d" % i, x_source, s, 11)
        x = F.conv_transpose1d(x, wn(sd, "ups.%d" % i), sd["ups.%d.bias" % i], stride=u, padding=u // 2 + u % 2,
                               output_padding=u % 2)
        x = x + x_source
        xs = None
        for j in range(nk):
            r = adain_resblock1(sd, "resblocks.%d" % (i * nk + j), x, s, rks[j], tuple(rds[j]))
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps["stage%d" % i] = x
    x = snake(x, sd["alphas.%d" % nu])
    x = F.conv1d(x, wn(sd, "conv_post"), sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def decoder(sd, cfg, asr, F0_curve, N, s, noise=None, har=None, taps=None):
    """Decoder.forward (eval), Modules/istftnet.py:499-528 / Modules/hifigan.py:446-475.
    cfg = config['model_params']['decoder']."""
    F0 = F.conv1d(F0_curve.unsqueeze(1), wn(sd, "F0_conv"), sd["F0_conv.bias"], stride=2, padding=1)
    Nn = F.conv1d(N.unsqueeze(1), wn(sd, "N_conv"), sd["N_conv.bias"], stride=2, padding=1)
    x = torch.cat([asr, F0, Nn], dim=1)
    x = adain_resblk1d(sd, "encode", x, s)
    if taps is not None:
        taps["encode"] = x
    asr_res = F.conv1d(asr, wn(sd, "asr_res.0"), sd["asr_res.0.bias"])
    res = True
    for i in range(4):
        if res:
            x = torch.cat([x, asr_res, F0, Nn], dim=1)
        x = adain_resblk1d(sd, "decode.%d" % i, x, s)
        if ("decode.%d.pool.weight_g" % i) in sd:
            res = False
    if taps is not None:
        taps["front"] = x
    gsd = sub(sd, "generator")
    gen = generator_istftnet if cfg["type"] == "istftnet" else generator_hifigan
    return gen(gsd, cfg, x, s, F0_curve, noise=noise, har=har, taps=taps)
