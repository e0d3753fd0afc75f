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


# ------------------------------------------------------------------------------------------------
# style-diffusion sampler (Modules/diffusion/{sampler,modules}.py)
# ------------------------------------------------------------------------------------------------
def _ada_layer_norm(sd, prefix, x, s):
    """AdaLayerNorm.forward, Modules/diffusion/modules.py:26-38 (x [B,N,C], s [B,style])."""
    h = F.linear(s, sd[prefix + ".fc.weight"], sd[prefix + ".fc.bias"])
    gamma, beta = torch.chunk(h.unsqueeze(1), 2, dim=-1)
    C = x.shape[-1]
    return (1 + gamma) * F.layer_norm(x, (C,), eps=1e-5) + beta


def _attention_base(sd, prefix, q, k, v, heads):
    """AttentionBase.forward, Modules/diffusion/modules.py:523-535 (no mask, no rel-pos)."""
    B, N, HD = q.shape
    D = HD // heads
    qh, kh, vh = (t.reshape(B, -1, heads, D).transpose(1, 2) for t in (q, k, v))
    sim = torch.einsum("bhnd,bhmd->bhnm", qh, kh) * (D ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhnm,bhmd->bhnd", attn, vh).transpose(1, 2).reshape(B, N, HD)
    return F.linear(out, sd[prefix + ".to_out.weight"], sd[prefix + ".to_out.bias"])


def _transformer_block(sd, prefix, x, heads, features=None):
    """TransformerBlock / StyleTransformerBlock.forward, modules.py:630-635 / 230-235 (self-attention only)."""
    a = prefix + ".attention"
    if features is None:  # Attention.forward, modules.py:575-584 (nn.LayerNorm)
        C = x.shape[-1]
        xn = F.layer_norm(x, (C,), sd[a + ".norm.weight"], sd[a + ".norm.bias"], 1e-5)
        cn = F.layer_norm(x, (C,), sd[a + ".norm_context.weight"], sd[a + ".norm_context.bias"], 1e-5)
    else:  # StyleAttention.forward, modules.py:269-281
        xn = _ada_layer_norm(sd, a + ".norm", x, features)
        cn = _ada_layer_norm(sd, a + ".norm_context", x, features)
    q = F.linear(xn, sd[a + ".to_q.weight"])
    k, v = torch.chunk(F.linear(cn, sd[a + ".to_kv.weight"]), 2, dim=-1)
    x = _attention_base(sd, a + ".attention", q, k, v, heads) + x
    f = prefix + ".feed_forward"
    y = F.linear(F.gelu(F.linear(x, sd[f + ".0.weight"], sd[f + ".0.bias"])), sd[f + ".2.weight"], sd[f + ".2.bias"])
    return y + x


def denoiser_net(sd, x, time, embedding, features=None, embedding_scale=1.0, heads=8, num_layers=3):
    """Transformer1d.forward (modules.py:402-425) / StyleTransformer1d.forward (modules.py:160-183).
    `sd` holds the net's own keys (the `diffusion.net.` / `unet.` prefix stripped).  multispeaker <=> features."""
    multispeaker = "to_features.0.weight" in sd

    def mapping():
        t = time.reshape(-1, 1)
        freqs = t * sd["to_time.0.0.weights"].reshape(1, -1) * 2 * math.pi  # modules.py:666-671
        four = torch.cat([t, freqs.sin(), freqs.cos()], dim=-1)
        m = F.gelu(F.linear(four, sd["to_time.0.1.weight"], sd["to_time.0.1.bias"]))
        if multispeaker:
            m = m + F.gelu(F.linear(features, sd["to_features.0.weight"], sd["to_features.0.bias"]))
        m = F.gelu(F.linear(m, sd["to_mapping.0.weight"], sd["to_mapping.0.bias"]))
        return F.gelu(F.linear(m, sd["to_mapping.2.weight"], sd["to_mapping.2.bias"]))

    def run(emb):
        m = mapping().unsqueeze(1)
        h = torch.cat([x.expand(-1, emb.size(1), -1), emb], dim=-1)
        for i in range(num_layers):
            h = h + m
            h = _transformer_block(sd, "blocks.%d" % i, h, heads, features if multispeaker else None)
        h = h.mean(dim=1).unsqueeze(1)  # no mask: padded positions are averaged too (modules.py:155,397)
        h = F.conv1d(h.transpose(1, 2), sd["to_out.1.weight"], sd["to_out.1.bias"])
        return h.transpose(-1, -2)

    if embedding_scale != 1.0:  # classifier-free guidance, modules.py:418-423
        N = embedding.shape[1]
        fixed = sd["fixed_embedding.embedding.weight"][:N].unsqueeze(0).expand(embedding.shape[0], -1, -1)
        out, out_masked = run(embedding), run(fixed)
        return out_masked + (out - out_masked) * embedding_scale
    return run(embedding)


def kdiffusion_denoise(sd, x_noisy, sigma, sigma_data, **kw):
    """KDiffusion.denoise_fn + get_scale_weights, Modules/diffusion/sampler.py:184-208."""
    B = x_noisy.shape[0]
    sigmas = torch.full((B,), float(sigma), dtype=torch.float32) if not torch.is_tensor(sigma) else \
        sigma.reshape(1).expand(B).float()
    c_noise = torch.log(sigmas) * 0.25
    sg = sigmas.reshape(B, 1, 1)
    c_skip = (sigma_data ** 2) / (sg ** 2 + sigma_data ** 2)
    c_out = sg * sigma_data * (sigma_data ** 2 + sg ** 2) ** -0.5
    c_in = (sg ** 2 + sigma_data ** 2) ** -0.5
    x_pred = denoiser_net(sd, c_in * x_noisy, c_noise, **kw)
    return c_skip * x_noisy + c_out * x_pred


def karras_schedule(num_steps, sigma_min=1e-4, sigma_max=3.0, rho=9.0):
    """KarrasSchedule.forward, sampler.py:328-337."""
    rho_inv = 1.0 / rho
    steps = torch.arange(num_steps, dtype=torch.float32)
    sigmas = (sigma_max ** rho_inv + (steps / (num_steps - 1)) * (sigma_min ** rho_inv - sigma_max ** rho_inv)) ** rho
    return F.pad(sigmas, pad=(0, 1), value=0.0)


def adpm2_sigmas(sigma, sigma_next):
    """ADPM2Sampler.get_sigmas with rho=1, sampler.py:490-495 (math.sqrt on 0-dim fp32 tensors -> python floats)."""
    sigma_up = math.sqrt(sigma_next ** 2 * (sigma ** 2 - sigma_next ** 2) / sigma ** 2)
    sigma_down = math.sqrt(sigma_next ** 2 - sigma_up ** 2)
    sigma_mid = ((sigma ** 1.0 + sigma_down ** 1.0) / 2) ** 1.0
    return sigma_up, sigma_down, sigma_mid


def sample_style(sd, noise, embedding, num_steps, step_noise, sigma_data=0.2, features=None, embedding_scale=1.0,
                 taps=None):
    """DiffusionSampler.forward + ADPM2Sampler.forward/step (sampler.py:573-586, 497-519), clamp=False.
    `step_noise` [num_steps-1, B, 1, C] replays the per-step randn_like draws (sampler.py:509)."""
    sigmas = karras_schedule(num_steps)
    kw = dict(embedding=embedding, features=features, embedding_scale=embedding_scale)
    fn = lambda xx, sg: kdiffusion_denoise(sd, xx, sg, sigma_data, **kw)
    x = sigmas[0] * noise
    for i in range(num_steps - 1):
        sigma, sigma_next = sigmas[i], sigmas[i + 1]
        s_up, s_down, s_mid = adpm2_sigmas(sigma, sigma_next)
        d = (x - fn(x, sigma)) / sigma
        x_mid = x + d * (s_mid - sigma)
        d_mid = (x_mid - fn(x_mid, s_mid)) / s_mid
        x = x + d_mid * (s_down - sigma)
        x = x + step_noise[i] * s_up
        if taps is not None:
            taps["step%d" % i] = x
    return x


# ------------------------------------------------------------------------------------------------
# text encoder, PL-BERT, prosody predictor, and the notebook glue
# ------------------------------------------------------------------------------------------------
def _lstm(sd, prefix, x, lengths=None):
    """nn.LSTM(bidirectional, batch_first) with pack/pad (models.py:314-327,545-566)."""
    w_ih = sd[prefix + ".weight_ih_l0"]
    lstm = torch.nn.LSTM(w_ih.shape[1], w_ih.shape[0] // 4, 1, batch_first=True, bidirectional=True)
    lstm.load_state_dict(sub(sd, prefix))
    if lengths is None or bool((lengths == x.shape[1]).all()):
        return lstm(x)[0]
    total = x.shape[1]
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=False)
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(packed)[0], batch_first=True, total_length=total)
    return y


def text_encoder(sd, tokens, lengths, mask):
    """TextEncoder.forward, models.py:302-331."""
    x = F.embedding(tokens, sd["embedding.weight"]).transpose(1, 2)
    m = mask.unsqueeze(1)
    x = x.masked_fill(m, 0.0)
    i = 0
    while ("cnn.%d.0.weight_g" % i) in sd:
        x = F.conv1d(x, wn(sd, "cnn.%d.0" % i), sd["cnn.%d.0.bias" % i], padding=2)
        x = F.layer_norm(x.transpose(1, -1), (x.shape[1],), sd["cnn.%d.1.gamma" % i], sd["cnn.%d.1.beta" % i],
                         1e-5).transpose(1, -1)
        x = F.leaky_relu(x, 0.2)
        x = x.masked_fill(m, 0.0)
        i += 1
    y = _lstm(sd, "lstm", x.transpose(1, 2), lengths).transpose(-1, -2)
    return y.masked_fill(m, 0.0)


def duration_encoder(sd, x, style, lengths, mask):
    """DurationEncoder.forward, models.py:536-569 (sd = predictor.text_encoder.*)."""
    N = x.shape[2]
    s = style.unsqueeze(1).expand(-1, N, -1)
    h = torch.cat([x.transpose(1, 2), s], dim=-1).masked_fill(mask.unsqueeze(-1), 0.0)
    i = 0
    while ("lstms.%d.weight_ih_l0" % i) in sd:
        h = _lstm(sd, "lstms.%d" % i, h, lengths)
        fc = F.linear(style, sd["lstms.%d.fc.weight" % (i + 1)], sd["lstms.%d.fc.bias" % (i + 1)])
        gamma, beta = torch.chunk(fc.unsqueeze(1), 2, dim=-1)
        h = (1 + gamma) * F.layer_norm(h, (h.shape[-1],), eps=1e-5) + beta
        h = torch.cat([h, s], dim=-1).masked_fill(mask.unsqueeze(-1), 0.0)
        i += 2
    return h


def f0n_train(sd, x, s):
    """ProsodyPredictor.F0Ntrain, models.py:497-510 (sd = predictor.*)."""
    y = _lstm(sd, "shared", x.transpose(-1, -2)).transpose(-1, -2)
    outs = []
    for name in ("F0", "N"):
        t = y
        for i in range(3):
            t = adain_resblk1d(sd, "%s.%d" % (name, i), t, s)
        outs.append(F.conv1d(t, sd[name + "_proj.weight"], sd[name + "_proj.bias"]).squeeze(1))
    return outs[0], outs[1]


def plbert(sd, plbert_params, tokens, attention_mask):
    """CustomAlbert.forward, Utils/PLBERT/util.py:6-12 (HF transformers AlbertModel; third-party, unpinned)."""
    from transformers import AlbertConfig, AlbertModel
    m = AlbertModel(AlbertConfig(**plbert_params)).eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        return m(tokens, attention_mask=attention_mask).last_hidden_state


def front(sds, cfg, plbert_params, tokens, lengths, noise, step_noise, diffusion_steps=5, embedding_scale=1.0,
          ref_s=None, alpha=0.3, beta=0.7, durations=None, taps=None, s_prev=None, t=0.7, lj_tail=None):
    """Everything the notebook `inference` cell does in front of the decoder call
    (Demo/Inference_LJSpeech.ipynb:268-311; Demo/Inference_LibriTTS.ipynb:258-322): returns the decoder's inputs
    (asr, F0, N, ref) and fills `taps`.  Batched over utterances of equal frame count."""
    B, N = tokens.shape
    mask = torch.gt(torch.arange(N).unsqueeze(0).expand(B, -1) + 1, lengths.unsqueeze(1))  # utils.py:42-46
    multispeaker = ref_s is not None
    t_en = text_encoder(sds["text_encoder"], tokens, lengths, mask)
    bert_dur = plbert(sds["bert"], plbert_params, tokens, (~mask).int())
    d_en = F.linear(bert_dur, sds["bert_encoder"]["weight"], sds["bert_encoder"]["bias"]).transpose(-1, -2)
    skw = dict(sigma_data=cfg["diffusion"]["dist"]["sigma_data"], embedding_scale=embedding_scale)
    if bool((lengths == N).all()):
        s_pred = sample_style(sub(sds["diffusion"], "unet"), noise, bert_dur, diffusion_steps, step_noise,
                              features=ref_s, **skw).squeeze(1)
    else:  # a right-padded batch: the notebooks synthesise ONE utterance per call, so the denoiser never sees padding
        s_pred = torch.cat([sample_style(sub(sds["diffusion"], "unet"), noise[b:b + 1],
                                         bert_dur[b:b + 1, :int(lengths[b])], diffusion_steps, step_noise[:, b:b + 1],
                                         features=None if ref_s is None else ref_s[b:b + 1], **skw).squeeze(1)
                            for b in range(B)])
    if s_prev is not None:  # LFinference: convex combination of previous and current style
        s_pred = t * s_prev + (1 - t) * s_pred
    s, ref = s_pred[:, 128:], s_pred[:, :128]
    if multispeaker:
        ref = alpha * ref + (1 - alpha) * ref_s[:, :128]
        s = beta * s + (1 - beta) * ref_s[:, 128:]
    if taps is not None:
        taps["s_mixed"] = torch.cat([ref, s], dim=-1)
    psd = sds["predictor"]
    d = duration_encoder(sub(psd, "text_encoder"), d_en, s, lengths, mask)
    if durations is None:
        x = _lstm(psd, "lstm", d, lengths)  # per-utterance semantics for a padded batch (pack/pad = un-padded run)
        dur = torch.sigmoid(F.linear(x, psd["duration_proj.linear_layer.weight"],
                                     psd["duration_proj.linear_layer.bias"])).sum(dim=-1)
        durations = torch.round(dur).clamp(min=1).long().masked_fill(mask, 0)
        if (not multispeaker) if lj_tail is None else lj_tail:
            durations[torch.arange(B), lengths - 1] += 5  # ipynb:301 `pred_dur[-1] += 5` of each utterance
    tot = durations.sum(dim=1)
    assert bool((tot == tot[0]).all()), "oracle.front batches utterances of equal frame count only: %s" % tot.tolist()
    T = int(tot[0])
    aln = torch.zeros(B, N, T)
    for b in range(B):  # the notebook's one-hot alignment loop, ipynb:303-307
        c = 0
        for i in range(N):
            aln[b, i, c:c + int(durations[b, i])] = 1
            c += int(durations[b, i])
    en = d.transpose(-1, -2) @ aln
    asr = t_en @ aln
    if cfg["decoder"]["type"] == "hifigan":
        en = torch.cat([en[:, :, :1], en[:, :, :-1]], dim=2)
        asr = torch.cat([asr[:, :, :1], asr[:, :, :-1]], dim=2)
    F0_pred, N_pred = f0n_train(psd, en, s)
    if taps is not None:
        taps.update(s_pred=s_pred, durations=durations, F0=F0_pred, N=N_pred, asr=asr,
                    en=en, t_en=t_en, d=d, bert_dur=bert_dur)
    return asr, F0_pred, N_pred, ref


def inference(sds, cfg, plbert_params, tokens, lengths, noise, step_noise, sine_noise, diffusion_steps=5,
              embedding_scale=1.0, ref_s=None, alpha=0.3, beta=0.7, durations=None, taps=None, s_prev=None, t=0.7,
              lj_tail=None):
    """The notebook `inference` cell (Demo/Inference_LJSpeech.ipynb:268-315; Demo/Inference_LibriTTS.ipynb:258-325),
    batched over equal-length utterances.  `sds` maps module name -> reference-layout state_dict; cfg =
    config['model_params']."""
    asr, F0_pred, N_pred, ref = front(sds, cfg, plbert_params, tokens, lengths, noise, step_noise,
                                      diffusion_steps=diffusion_steps, embedding_scale=embedding_scale, ref_s=ref_s,
                                      alpha=alpha, beta=beta, durations=durations, taps=taps, s_prev=s_prev, t=t,
                                      lj_tail=lj_tail)
    return decoder(sds["decoder"], cfg["decoder"], asr, F0_pred, N_pred, ref, noise=sine_noise, taps=taps)


def long_form(sds, cfg, plbert_params, sentences, noises, step_noises, sine_noises, diffusion_steps=5,
              embedding_scale=1.0, ref_s=None, alpha=0.3, beta=0.7, t=0.7, trim=None):
    """The long-form driver loops of the notebooks (Demo/Inference_LibriTTS.ipynb LFinference + "for text in
    sentences"; Demo/Inference_LJSpeech.ipynb "Long-form generation"): sentence k is synthesised with
    s_prev = the mixed style LFinference returned for sentence k-1; no +5 tail frames; `trim` samples cut from each
    sentence's end (100 in the LibriTTS notebook, none in the LJSpeech one)."""
    if trim is None:
        trim = 100 if ref_s is not None else 0
    s_prev, waves = None, []
    for k, tok in enumerate(sentences):
        taps = {}
        tokens = tok.reshape(1, -1)
        w = inference(sds, cfg, plbert_params, tokens, torch.LongTensor([tokens.shape[1]]), noises[k], step_noises[k],
                      sine_noises[k], diffusion_steps=diffusion_steps, embedding_scale=embedding_scale, ref_s=ref_s,
                      alpha=alpha, beta=beta, taps=taps, s_prev=s_prev, t=t, lj_tail=False)
        s_prev = taps["s_mixed"]
        w = w.reshape(-1)
        waves.append(w[:-trim] if trim else w)
    return waves, s_prev


# ------------------------------------------------------------------------------------------------
# reference-audio style path: StyleEncoder (models.py:139-164) over ResBlk (:97-137), LearnedDownSample (:27-42),
# DownSample (:63-77), all under old-style torch.nn.utils.spectral_norm (state_dict: weight_orig / weight_u / weight_v)
# ------------------------------------------------------------------------------------------------
def sn_weight(sd, prefix):
    """Eval-mode spectral norm: no power iteration, W = weight_orig / sigma with sigma = u . (W_mat v)
    (torch/nn/utils/spectral_norm.py compute_weight, do_power_iteration=False)."""
    w = sd[prefix + ".weight_orig"].float()
    u, v = sd[prefix + ".weight_u"].float(), sd[prefix + ".weight_v"].float()
    return w / torch.dot(u, torch.mv(w.reshape(w.shape[0], -1), v))


def _style_resblk(sd, p, x):
    """ResBlk(normalize=False, downsample='half'), models.py:97-137."""
    sc = x
    if (p + ".conv1x1.weight_orig") in sd:                                  # learned_sc: dim_in != dim_out
        sc = F.conv2d(sc, sn_weight(sd, p + ".conv1x1"))
    if sc.shape[-1] % 2 != 0:                                               # DownSample('half'), models.py:72-75
        sc = torch.cat([sc, sc[..., -1].unsqueeze(-1)], dim=-1)
    sc = F.avg_pool2d(sc, 2)
    r = F.conv2d(F.leaky_relu(x, 0.2), sn_weight(sd, p + ".conv1"), sd[p + ".conv1.bias"], 1, 1)
    c = r.shape[1]                                                          # LearnedDownSample('half'): depthwise 3x3 / 2
    r = F.conv2d(r, sn_weight(sd, p + ".downsample_res.conv"), sd[p + ".downsample_res.conv.bias"], 2, 1, 1, c)
    r = F.conv2d(F.leaky_relu(r, 0.2), sn_weight(sd, p + ".conv2"), sd[p + ".conv2.bias"], 1, 1)
    return (sc + r) / math.sqrt(2)


def style_encoder(sd, mel):
    """StyleEncoder.forward (models.py:139-164): mel [B, 1, 80, T] -> style [B, style_dim]; `sd` is the module's own
    state_dict (keys `shared.N...`, `unshared...`)."""
    sd = {k: v.float() for k, v in sd.items()}
    h = F.conv2d(mel.float(), sn_weight(sd, "shared.0"), sd["shared.0.bias"], 1, 1)
    for i in range(1, 5):
        h = _style_resblk(sd, "shared.%d" % i, h)
    h = F.conv2d(F.leaky_relu(h, 0.2), sn_weight(sd, "shared.6"), sd["shared.6.bias"])   # 5x5, valid
    h = F.leaky_relu(F.adaptive_avg_pool2d(h, 1), 0.2).reshape(h.shape[0], -1)
    return F.linear(h, sd["unshared.weight"], sd["unshared.bias"])


def compute_style(sd_style, sd_pred, wave):
    """`compute_style` of Demo/Inference_LibriTTS.ipynb:100-111 minus the file I/O: wave [B, L] at 24 kHz -> ref_s
    [B, 256] = cat(style_encoder(mel), predictor_encoder(mel)); the mel is oracle/mel_ref.py (fp64) rounded to fp32."""
    from oracle import mel_ref
    mel = mel_ref.mel_spectrogram_t(wave).unsqueeze(1)
    return torch.cat([style_encoder(sd_style, mel), style_encoder(sd_pred, mel)], dim=1)
