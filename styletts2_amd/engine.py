"""Python binding of the module-level C ABI (include/st2.h: st2_create / st2_load_weights / st2_finalize_weights /
st2_decoder_forward / st2_sampler_run / st2_prosody_forward).

The launch plans of `Decoder.forward` (Modules/istftnet.py:499-528, Modules/hifigan.py:446-475) and
`DiffusionSampler.forward` (Modules/diffusion/sampler.py:573-586) live in C++ (csrc/st2_engine.hip); a module forward is
ONE ctypes call here: PyTorch supplies the device buffers (inputs, output, one workspace) and the stream, nothing else.
The per-kernel Python plans (decoder.py, diffusion.py, text.py, style.py) remain as the tap-point path of the parity
tests (`_hooks.override(plan="python")`, tests only -- no environment switch), and CPU tensors always take them (the CPU
plan tests substitute per-kernel contracts for the HIP wrappers; the real wrappers raise on a CPU tensor).
"""
import ctypes as C
import math
import weakref

import torch

from . import _hooks, _lib
from . import weights as W


def plan_mode():
    """"engine": module forwards are single C-ABI calls into the C++ plans (the product path); "python": the
    per-kernel Python plans, selectable from tests only (_hooks.py)."""
    return _hooks.plan


def _folded_state(module):
    """Reference-layout state_dict with every weight-norm pair folded: X.weight_g / X.weight_v -> X.weight."""
    sd = module.state_dict()
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            out[k[:-2]] = W.fold_weight_norm(sd[k[:-1] + "g"].detach().float().cpu(), v.detach().float().cpu())
        elif torch.is_floating_point(v):
            out[k] = v.detach().float().cpu()
    return out


_LIVE = weakref.WeakSet()  # every Engine of the process (pipeline.calibrate folds one recording into each of them)


def live_engines():
    return [e for e in _LIVE if getattr(e, "h", None)]


def norm_device(dev):
    """torch.device with its index filled in: 'cuda', 'cuda:0', torch.device('cuda') and a tensor's .device all name the
    same engine (advisor, round 5: a cache keyed on the caller's spelling rebuilt the engine -- a full repack and upload --
    and silently dropped its calibration table)."""
    if dev is None:
        return None
    dev = torch.device(dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def same_device(eng, dev):
    return eng is not None and norm_device(eng.device) == norm_device(dev)


def replaced(old, what):
    """Called when a cached engine is about to be rebuilt (weights reloaded or moved): a calibration table belongs to the
    weights it was measured on and does not travel -- but the caller must hear that it is gone."""
    if old is not None and getattr(old, "h", None) and any(old.calibration_scales()):
        import warnings
        warnings.warn("the calibrated %s engine is being rebuilt (weights reloaded or moved): its per-layer operand scales are "
                      "dropped; run pipeline.calibrate / load_calibration_state again" % what, RuntimeWarning, stacklevel=3)


class Engine:
    """One st2_engine handle (packed weights of a decoder and / or a denoiser on the current device)."""

    def __init__(self, cfg):
        self.lib = _lib.load()
        self.cfg = cfg
        self.h = C.c_void_p()
        _lib.check(self.lib.st2_create(C.byref(cfg), C.byref(self.h)), "st2_create")
        self.device = None
        self.calib_gen = 0  # bumped whenever the per-layer operand scales change: recorded graphs hold the old ones
        _LIVE.add(self)

    # -- per-layer operand scales of the split-f16 convs (include/st2.h st2_calibrate) -------------------------------------
    def calibrate(self, margin_bits=3):
        """Folds the headroom records of the forwards just run under `ops.headroom()` into this engine's per-conv x_scale
        table.  Returns (sites set, launches that ran into the f16 clamp at their old scale: > 0 = record and call again)."""
        clamped = C.c_int32(0)
        n = self.lib.st2_calibrate(self.h, int(margin_bits), C.byref(clamped))
        if n < 0:
            _lib.check(1, "st2_calibrate")
        if n > 0:
            self.calib_gen += 1
        return n, int(clamped.value)

    def calibration(self):
        """[{site, name, C_in, C_out, ks, x_scale (0.0 = by rule), seen (max |pro(x)| at calibration)}] for every conv site."""
        n = self.lib.st2_calibration_read(self.h, None, 0)
        W = _lib.CALIBRATION_COLS
        buf = (C.c_double * (W * max(n, 1)))()
        n = min(n, self.lib.st2_calibration_read(self.h, buf, n))
        name = C.create_string_buffer(256)
        out = []
        for i in range(max(n, 0)):
            _lib.check(self.lib.st2_calibration_site_name(self.h, i, name, len(name)), "st2_calibration_site_name")
            r = buf[W * i:W * i + W]
            out.append({"site": i, "name": name.value.decode(), "C_in": int(r[0]), "C_out": int(r[1]), "ks": int(r[2]),
                        "x_scale": r[3], "seen": r[4]})
        return out

    def calibration_scales(self):
        return [r["x_scale"] for r in self.calibration()]

    def set_calibration(self, scales):
        """Installs a table (one power of two or 0.0 = rule per site, as `calibration_scales()` returned it on a process
        holding the same model), or clears it (None / empty)."""
        scales = list(scales or [])
        arr = (C.c_float * max(len(scales), 1))(*scales)
        _lib.check(self.lib.st2_calibration_write(self.h, arr, len(scales)), "st2_calibration_write")
        self.calib_gen += 1

    def __del__(self):
        try:
            if getattr(self, "h", None) is not None and self.h:
                self.lib.st2_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def load(self, name, t):
        t = t.detach().float().cpu().contiguous()
        shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        _lib.check(self.lib.st2_load_weights(self.h, name.encode(), t.data_ptr(), shape, t.dim()), "st2_load_weights")

    def load_module(self, prefix, module):
        for k, v in _folded_state(module).items():
            self.load(prefix + k, v)

    def finalize(self, which, device):
        if device is not None and torch.device(device).type == "cuda":
            with torch.cuda.device(device):
                _lib.check(self.lib.st2_finalize_weights(self.h, which), "st2_finalize_weights")
        else:
            _lib.check(self.lib.st2_finalize_weights(self.h, which), "st2_finalize_weights")
        self.device = device

    # -- decoder ---------------------------------------------------------------------------------------------------
    def decoder_forward(self, asr, F0, N, s, noise=None, har=None, taps=None):
        B, Cin, T = asr.shape
        dev = asr.device
        cfg = self.cfg
        rates = [cfg.upsample_rates[i] for i in range(cfg.n_upsamples)]
        up = math.prod(rates) * (cfg.gen_istft_hop if cfg.decoder_kind == 0 else 1)
        L = 2 * T * up
        asr, F0, N, s = (t.float().contiguous() for t in (asr, F0, N, s))
        if har is None and noise is None:
            noise = torch.randn(B, L, 9, device=dev, dtype=torch.float32)  # the reference's in-forward randn_like
        if noise is not None:
            noise = noise.float().contiguous()
            assert noise.shape == (B, L, 9)
        if har is not None:
            har = har.float().contiguous()
        wave = torch.empty((B, 1, L), device=dev, dtype=torch.float32)
        nbytes = self.lib.st2_decoder_workspace_bytes(self.h, B, T)
        if nbytes <= 0:
            raise _lib.St2Error("st2_decoder_workspace_bytes failed (weights not finalized?)")
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        tp = None
        if taps is not None:
            tp = _lib.DecoderTaps()
            M = L // cfg.gen_istft_hop + 1 if cfg.decoder_kind == 0 else L
            bufs = {"encode": torch.empty((B, 1024, T), device=dev), "front": torch.empty((B, 512, 2 * T), device=dev)}
            if har is None:
                bufs["har_source"] = torch.empty((B, L), device=dev)
            if cfg.decoder_kind == 0:
                bufs["har"] = torch.empty((B, cfg.gen_istft_n_fft + 2, M), device=dev)
                bufs["spec_phase"] = torch.empty((B, cfg.gen_istft_n_fft + 2, M), device=dev)
            Ls = 2 * T
            for i in range(cfg.n_upsamples):  # stage lengths as decoder_plan computes them
                u, k = rates[i], cfg.upsample_kernel_sizes[i]
                if cfg.decoder_kind == 0:
                    Ls = (Ls - 1) * u - 2 * ((k - u) // 2) + k + (1 if i + 1 == cfg.n_upsamples else 0)
                else:
                    Ls = (Ls - 1) * u - 2 * (u // 2 + u % 2) + k + u % 2
                bufs["stage%d" % i] = torch.empty((B, cfg.upsample_initial_channel >> (i + 1), Ls), device=dev)
            for k, v in bufs.items():
                if k.startswith("stage"):
                    tp.stage[int(k[5:])] = v.data_ptr()
                else:
                    setattr(tp, k, v.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_decoder_forward(self.h, asr.data_ptr(), F0.data_ptr(), N.data_ptr(), s.data_ptr(),
                                                0 if noise is None else noise.data_ptr(),
                                                0 if har is None else har.data_ptr(), B, T, wave.data_ptr(),
                                                ws_ptr, nbytes, None if tp is None else C.byref(tp), stream),
                   "st2_decoder_forward")
        if taps is not None:
            taps.update(bufs)
            if cfg.decoder_kind == 0:
                taps["har"] = bufs["har"] if har is None else har
            else:
                taps["har"] = (bufs["har_source"] if har is None else har.reshape(B, L)).unsqueeze(1)
        return wave

    # -- prosody (alignment expansion + F0Ntrain) --------------------------------------------------------------------
    def prosody_forward(self, d_cm, t_en, durations, s, T, shift=False):
        """d_cm [B, d_hid + sty, N], t_en [B, dim_in, N], durations int64 [B, N] (rows summing to T), s [B, sty] ->
        (asr [B, dim_in, T], F0 [B, 2T], N [B, 2T]): one `st2_prosody_forward` call."""
        B, Cd, N = d_cm.shape
        dev = d_cm.device
        d_cm, t_en, s = (t.float().contiguous() for t in (d_cm, t_en, s))
        durations = durations.long().contiguous()
        assert t_en.shape == (B, self.cfg.dim_in, N) and durations.shape == (B, N)
        assert Cd == self.cfg.pred_hidden + self.cfg.style_dim and s.shape == (B, self.cfg.style_dim)
        asr = torch.empty((B, self.cfg.dim_in, T), device=dev, dtype=torch.float32)
        f0 = torch.empty((B, 2 * T), device=dev, dtype=torch.float32)
        nn_ = torch.empty((B, 2 * T), device=dev, dtype=torch.float32)
        nbytes = self.lib.st2_prosody_workspace_bytes(self.h, B, N, T)
        if nbytes <= 0:
            raise _lib.St2Error("st2_prosody_workspace_bytes failed (predictor weights not finalized?)")
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_prosody_forward(self.h, d_cm.data_ptr(), t_en.data_ptr(), durations.data_ptr(), s.data_ptr(),
                                                B, N, T, 1 if shift else 0, asr.data_ptr(), f0.data_ptr(), nn_.data_ptr(),
                                                ws_ptr, nbytes, stream), "st2_prosody_forward")
        return asr, f0, nn_

    # -- text encoder --------------------------------------------------------------------------------------------------
    def text_forward(self, tokens, lengths=None):
        """tokens int64 [B, N], lengths int32 [B] on the device or None -> t_en [B, dim_in, N]: one `st2_text_forward` call."""
        B, N = tokens.shape
        dev = tokens.device
        tokens = tokens.long().contiguous()
        if lengths is not None:
            lengths = lengths.to(torch.int32).contiguous()
            assert lengths.device == dev and lengths.numel() == B
        t_en = torch.empty((B, self.cfg.dim_in, N), device=dev, dtype=torch.float32)
        nbytes = self.lib.st2_text_workspace_bytes(self.h, B, N)
        if nbytes <= 0:
            raise _lib.St2Error("st2_text_workspace_bytes failed (text-encoder weights not finalized?)")
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_text_forward(self.h, tokens.data_ptr(), 0 if lengths is None else lengths.data_ptr(), B, N,
                                             t_en.data_ptr(), ws_ptr, nbytes, stream), "st2_text_forward")
        return t_en

    # -- reference-audio style encoders ---------------------------------------------------------------------------------------
    def style_forward(self, which, mel):
        """mel [B, 1, 80, T] or [B, 80, T] (normalised log-mel, T >= 80) -> [B, style_dim]: one `st2_style_forward` call;
        which = 0 `style_encoder`, 1 `predictor_encoder`."""
        mel = mel.float().reshape(mel.shape[0], mel.shape[-2], mel.shape[-1]).contiguous()
        B, H, T = mel.shape
        dev = mel.device
        nbytes = self.lib.st2_style_workspace_bytes(self.h, which, B, H, T)
        if nbytes <= 0:
            raise _lib.St2Error("st2_style_workspace_bytes failed (style-encoder weights not finalized, or not an 80 x >= 80 mel)")
        out = torch.empty((B, self.style_dims[which]), device=dev, dtype=torch.float32)
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_style_forward(self.h, which, mel.data_ptr(), B, H, T, out.data_ptr(), ws_ptr, nbytes, stream),
                   "st2_style_forward")
        return out

    # -- PL-BERT ---------------------------------------------------------------------------------------------------------
    def bert_forward(self, tokens, lengths=None):
        """tokens int64 [B, N], lengths int32 [B] on the device or None -> last hidden state [B, N, hidden] (a transposed
        view of the channel-major buffer `st2_bert_forward` fills)."""
        B, N = tokens.shape
        dev = tokens.device
        tokens = tokens.long().contiguous()
        if lengths is not None:
            lengths = lengths.to(torch.int32).contiguous()
            assert lengths.device == dev and lengths.numel() == B
        out = torch.empty((B, self.cfg.dn_embedding, N), device=dev, dtype=torch.float32)
        nbytes = self.lib.st2_bert_workspace_bytes(self.h, B, N)
        if nbytes <= 0:
            raise _lib.St2Error("st2_bert_workspace_bytes failed (PL-BERT weights not finalized?)")
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_bert_forward(self.h, tokens.data_ptr(), 0 if lengths is None else lengths.data_ptr(), B, N,
                                             out.data_ptr(), ws_ptr, nbytes, stream), "st2_bert_forward")
        return out.transpose(1, 2)

    # -- the whole front (tokens -> t_en, d, s, ref, durations) ---------------------------------------------------------------
    def front_forward(self, tokens, noise, step_noise, table, sigma0, *, lengths=None, ref_s=None, s_prev=None,
                      embedding_scale=1.0, alpha=0.3, beta=0.7, t=0.7, predict=True, tail=0, carry=False):
        """One `st2_front_forward` call (== pipeline._front_core): tokens int64 [B, N], noise [B, 1, 256] or [B, 256],
        step_noise [steps-1, B, 1, 256]; (table, sigma0) = DiffusionSampler.step_table(steps).  Returns a dict with t_en
        [B, dim_in, N], d_cm [B, d_hid + sty, N], s, ref [B, sty], s_pred [B, 2 sty] (ref | s) and durations int64 [B, N]
        (None unless `predict`).  `carry`: the rows are consecutive sentences of one passage, row k's style is mixed with row
        k-1's mixed style (`s_prev` [1, 2 sty] or None feeds row 0): st2.h st2_front_args.carry."""
        cfg = self.cfg
        B, N = tokens.shape
        dev = tokens.device
        C2 = cfg.dn_channels
        steps = step_noise.shape[0] + 1
        f = lambda v: None if v is None else v.float().contiguous()
        tokens = tokens.long().contiguous()
        noise, step_noise, ref_s, s_prev = f(noise), f(step_noise), f(ref_s), f(s_prev)
        assert noise.numel() == B * C2 and step_noise.numel() == (steps - 1) * B * C2 and len(table) == (steps - 1) * 11
        assert s_prev is None or s_prev.numel() == (C2 if carry else B * C2)
        if lengths is not None:
            lengths = lengths.to(torch.int32).contiguous()
            assert lengths.device == dev and lengths.numel() == B
        new = lambda *shape, dtype=torch.float32: torch.empty(shape, device=dev, dtype=dtype)
        out = dict(t_en=new(B, cfg.dim_in, N), d_cm=new(B, cfg.pred_hidden + cfg.style_dim, N), s=new(B, cfg.style_dim),
                   ref=new(B, cfg.style_dim), s_pred=new(B, C2),
                   durations=new(B, N, dtype=torch.int64) if predict else None)
        ptr = lambda v: None if v is None else v.data_ptr()
        tab = (C.c_double * len(table))(*table)
        a = _lib.FrontArgs(tokens=ptr(tokens), lengths=ptr(lengths), noise=ptr(noise), step_noise=ptr(step_noise),
                           ref_s=ptr(ref_s), s_prev=ptr(s_prev), B=B, N=N, steps=steps, tail=int(tail),
                           embedding_scale=float(embedding_scale), table=tab, sigma0=float(sigma0), alpha=float(alpha),
                           beta=float(beta), t=float(t), t_en=ptr(out["t_en"]), d_cm=ptr(out["d_cm"]), s=ptr(out["s"]),
                           ref=ptr(out["ref"]), s_pred_out=ptr(out["s_pred"]), durations=ptr(out["durations"]), carry=int(bool(carry)))
        nbytes = self.lib.st2_front_workspace_bytes(self.h, C.byref(a))
        if nbytes <= 0:
            raise _lib.St2Error("st2_front_workspace_bytes failed (a weight group is not finalized?)")
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_front_forward(self.h, C.byref(a), ws_ptr, nbytes, stream), "st2_front_forward")
        return out

    # -- duration stage (DurationEncoder + duration LSTM + head) -------------------------------------------------------
    def duration_forward(self, d_en, s, lengths=None, tail=0, want_durations=True):
        """d_en [B, d_hid, N] (bert_encoder output, channel-major), s [B, sty], lengths int32 [B] on the device or None ->
        (d_cm [B, d_hid + sty, N], durations int64 [B, N] or None): one `st2_duration_forward` call."""
        B, dh, N = d_en.shape
        dev = d_en.device
        d_en, s = d_en.float().contiguous(), s.float().contiguous()
        assert dh == self.cfg.pred_hidden and s.shape == (B, self.cfg.style_dim)
        if lengths is not None:
            lengths = lengths.to(torch.int32).contiguous()
            assert lengths.device == dev and lengths.numel() == B
        d_cm = torch.empty((B, dh + self.cfg.style_dim, N), device=dev, dtype=torch.float32)
        dur = torch.empty((B, N), device=dev, dtype=torch.int64) if want_durations else None
        nbytes = self.lib.st2_duration_workspace_bytes(self.h, B, N)
        if nbytes <= 0:
            raise _lib.St2Error("st2_duration_workspace_bytes failed (duration-encoder weights not finalized?)")
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_duration_forward(self.h, d_en.data_ptr(), s.data_ptr(),
                                                 0 if lengths is None else lengths.data_ptr(), B, N, int(tail),
                                                 d_cm.data_ptr(), 0 if dur is None else dur.data_ptr(), ws_ptr, nbytes,
                                                 stream), "st2_duration_forward")
        return d_cm, dur

    # -- sampler ---------------------------------------------------------------------------------------------------
    def sampler_run(self, noise, embedding, features, step_noise, lengths, steps, scale, table, sigma0, taps=None):
        B = noise.shape[0]
        Cc = self.cfg.dn_channels
        N = embedding.shape[1]
        dev = noise.device
        noise = noise.float().contiguous()
        embedding = embedding.float().contiguous()
        if features is not None:
            features = features.float().contiguous()
        step_noise = step_noise.float().contiguous()
        assert step_noise.numel() == (steps - 1) * B * Cc and noise.numel() == B * Cc
        out = torch.empty((B, 1, Cc), device=dev, dtype=torch.float32)
        nbytes = self.lib.st2_sampler_workspace_bytes(self.h, B, N, steps, float(scale))
        if nbytes <= 0:
            raise _lib.St2Error("st2_sampler_workspace_bytes failed (weights not finalized?)")
        ws = torch.empty((nbytes + 256,), device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        st = torch.empty((steps - 1, B, 1, Cc), device=dev, dtype=torch.float32) if taps is not None else None
        tab = (C.c_double * len(table))(*table)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(self.lib.st2_sampler_run(self.h, noise.data_ptr(), embedding.data_ptr(),
                                            0 if features is None else features.data_ptr(), step_noise.data_ptr(),
                                            0 if lengths is None else lengths.data_ptr(), B, N, steps, float(scale), tab,
                                            float(sigma0), out.data_ptr(), ws_ptr, nbytes,
                                            0 if st is None else st.data_ptr(), stream), "st2_sampler_run")
        if taps is not None:
            for i in range(steps - 1):
                taps["step%d" % i] = st[i]
        return out


def predictor_config(pred, dim_in):
    """st2_model_config of a styletts2_amd.text.ProsodyPredictor (F0Ntrain only; decoder / denoiser fields are placeholders)."""
    cfg = _lib.ModelConfig()
    cfg.decoder_kind, cfg.dim_in, cfg.upsample_initial_channel = 0, dim_in, 512
    cfg.n_upsamples, cfg.n_resblock_kernels = 1, 1
    cfg.style_dim = pred.F0[0].norm1.fc.weight.shape[1]
    cfg.pred_hidden = pred.shared.hidden_size * 2
    return cfg


def build_predictor_engine(pred, device, dim_in=512):
    """Engine handle holding the whole prosody predictor: duration encoder + duration head (st2_duration_forward) and
    F0Ntrain (st2_prosody_forward)."""
    eng = Engine(predictor_config(pred, dim_in))
    sd = _folded_state(pred)
    for k, v in sd.items():
        if k.split(".")[0] in ("shared", "F0", "N", "F0_proj", "N_proj", "text_encoder", "lstm", "duration_proj"):
            eng.load("predictor." + k, v)
    eng.finalize(4, device)
    return eng


def build_text_engine(text_encoder, device):
    """Engine handle holding the text encoder (st2_text_forward)."""
    cfg = _lib.ModelConfig()
    cfg.decoder_kind, cfg.upsample_initial_channel, cfg.style_dim = 0, 512, 128
    cfg.n_upsamples, cfg.n_resblock_kernels = 1, 1
    cfg.dim_in = text_encoder.embedding.weight.shape[1]
    eng = Engine(cfg)
    eng.load_module("text_encoder.", text_encoder)
    eng.finalize(8, device)
    return eng


def _style_state(enc):
    """StyleEncoder parameters as st2_load_weights wants them: every spectral-norm triple folded on the host
    (X.weight = weight_orig / (u . (W_mat v)), style._SNConv2d.folded_host), plain tensors as they are."""
    from .style import _SNConv2d
    out = {}
    for name, m in enc.named_modules():
        if isinstance(m, _SNConv2d):
            out[name + ".weight"] = m.folded_host()
            if m.bias is not None:
                out[name + ".bias"] = m.bias.detach().float().cpu()
    out["unshared.weight"] = enc.unshared.weight.detach().float().cpu()
    out["unshared.bias"] = enc.unshared.bias.detach().float().cpu()
    return out


def build_style_engine(style_encoder, predictor_encoder, device):
    """Engine handle holding the two reference-audio style encoders (st2_style_forward; either may be None)."""
    cfg = _lib.ModelConfig()
    cfg.decoder_kind, cfg.dim_in, cfg.upsample_initial_channel, cfg.style_dim = 0, 512, 512, 128
    cfg.n_upsamples, cfg.n_resblock_kernels = 1, 1
    eng = Engine(cfg)
    eng.style_dims = [0, 0]
    for which, (prefix, enc) in enumerate((("style_encoder.", style_encoder), ("predictor_encoder.", predictor_encoder))):
        if enc is not None:
            for k, v in _style_state(enc).items():
                eng.load(prefix + k, v)
            eng.style_dims[which] = enc.unshared.out_features
    eng.finalize(32, device)
    return eng


def _bert_fields(cfg, bert):
    cfg.bert_layers, cfg.bert_ln_eps = bert.config.num_hidden_layers, bert.config.layer_norm_eps
    assert bert.config.num_hidden_groups == 1 and bert.config.inner_group_num == 1, "PL-BERT shares one ALBERT layer"
    assert bert.config.hidden_act == "gelu_new" and bert.config.hidden_size // bert.config.num_attention_heads == 64


def build_bert_engine(bert, device):
    """Engine handle holding PL-BERT (st2_bert_forward)."""
    cfg = _lib.ModelConfig()
    cfg.decoder_kind, cfg.dim_in, cfg.upsample_initial_channel, cfg.style_dim = 0, 512, 512, 128
    cfg.n_upsamples, cfg.n_resblock_kernels = 1, 1
    cfg.dn_embedding = bert.config.hidden_size
    _bert_fields(cfg, bert)
    eng = Engine(cfg)
    eng.load_module("bert.", bert)
    eng.finalize(16, device)
    return eng


def build_front_engine(model, device):
    """Engine handle holding everything in front of the alignment -- text encoder, PL-BERT + bert_encoder, style
    denoiser, prosody predictor (duration encoder / head and F0Ntrain): st2_front_forward, st2_prosody_forward."""
    net = model.diffusion.diffusion.net
    cfg = denoiser_config(net)
    pc = predictor_config(model.predictor, model.text_encoder.embedding.weight.shape[1])
    cfg.dim_in, cfg.style_dim, cfg.pred_hidden = pc.dim_in, pc.style_dim, pc.pred_hidden
    _bert_fields(cfg, model.bert)
    eng = Engine(cfg)
    eng.load_module("text_encoder.", model.text_encoder)
    eng.load_module("bert.", model.bert)
    eng.load_module("bert_encoder.", model.bert_encoder)
    eng.load_module("denoiser.", net)
    eng.load_module("predictor.", model.predictor)
    eng.finalize(2 | 4 | 8 | 16, device)
    return eng


def model_config(model):
    """st2_model_config of a whole styletts2_amd model dict (decoder + denoiser + predictor + text encoder + PL-BERT)."""
    cfg = decoder_config(model.decoder)
    dn = denoiser_config(model.diffusion.diffusion.net)
    for k in ("multispeaker", "dn_layers", "dn_heads", "dn_head_features", "dn_multiplier", "dn_channels", "dn_embedding",
              "dn_context_features", "dn_max_length"):
        setattr(cfg, k, getattr(dn, k))
    cfg.pred_hidden = model.predictor.shared.hidden_size * 2
    assert cfg.dim_in == model.text_encoder.embedding.weight.shape[1]
    _bert_fields(cfg, model.bert)
    return cfg


MODEL_PREFIXES = (("text_encoder.", "text_encoder"), ("bert.", "bert"), ("bert_encoder.", "bert_encoder"),
                  ("predictor.", "predictor"), ("decoder.", "decoder"))


def model_state(model):
    """name -> folded fp32 tensor of everything st2_load_weights wants for the text -> waveform path."""
    out = {}
    for prefix, key in MODEL_PREFIXES:
        for k, v in _folded_state(model[key]).items():
            out[prefix + k] = v
    for k, v in _folded_state(model.diffusion.diffusion.net).items():
        out["denoiser." + k] = v
    return out


def build_model_engine(model, device):
    """ONE engine handle for the whole text -> waveform path: st2_front_forward, st2_prosody_forward, st2_decoder_forward."""
    eng = Engine(model_config(model))
    for k, v in model_state(model).items():
        eng.load(k, v)
    eng.finalize(31, device)
    return eng


def decoder_config(dec):
    """st2_model_config of a styletts2_amd.decoder.Decoder (denoiser fields left zero)."""
    g = dec.generator
    cfg = _lib.ModelConfig()
    cfg.decoder_kind = 0 if dec.kind == "istftnet" else 1
    cfg.dim_in = dec.dim_in
    cfg.style_dim = dec.encode.norm1.fc.weight.shape[1]
    cfg.upsample_initial_channel = g.channels[0] * 2
    cfg.n_upsamples = g.num_upsamples
    for i in range(g.num_upsamples):
        cfg.upsample_rates[i] = g.rates[i]
        cfg.upsample_kernel_sizes[i] = g.up_ks[i]
    cfg.n_resblock_kernels = g.num_kernels
    for k in range(g.num_kernels):
        rb = g.resblocks[k]
        cfg.resblock_kernel_sizes[k] = rb.ks
        for j in range(3):
            cfg.resblock_dilations[k][j] = rb.dilation[j]
    if dec.kind == "istftnet":
        cfg.gen_istft_n_fft, cfg.gen_istft_hop = g.n_fft, g.hop
    return cfg


def denoiser_config(net):
    """st2_model_config of a styletts2_amd.diffusion._Transformer (decoder fields are placeholders)."""
    cfg = _lib.ModelConfig()
    cfg.decoder_kind, cfg.dim_in, cfg.style_dim, cfg.upsample_initial_channel = 0, 512, 128, 512
    cfg.n_upsamples, cfg.n_resblock_kernels = 1, 1
    cfg.multispeaker = 1 if net.multispeaker else 0
    cfg.dn_layers = len(net.blocks)
    cfg.dn_heads, cfg.dn_head_features = net.heads, net.head_features
    cfg.dn_multiplier = net.blocks[0].feed_forward[0].weight.shape[0] // net.features if len(net.blocks) else 0
    cfg.dn_channels, cfg.dn_embedding = net.channels, net.emb_features
    cfg.dn_context_features = net.to_features[0].weight.shape[1] if net.multispeaker else 0
    cfg.dn_max_length = net.fixed_embedding.max_length
    return cfg


def build_decoder_engine(dec, device):
    eng = Engine(decoder_config(dec))
    eng.load_module("decoder.", dec)
    eng.finalize(1, device)
    return eng


def build_denoiser_engine(net, device):
    eng = Engine(denoiser_config(net))
    eng.load_module("denoiser.", net)
    eng.finalize(2, device)
    return eng
