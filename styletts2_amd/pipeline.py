"""Text -> waveform inference glue: the logic that in the reference exists
only inside notebook cells (Demo/Inference_LJSpeech.ipynb:268-315, Demo/Inference_LibriTTS.ipynb:258-325),
batched over utterances and kept on the device.

Differences from the notebooks, all host-side:
  * a batch of B utterances instead of one;
  * the duration -> alignment step is an index gather (`expand_by_durations`) instead of a Python loop building a
    dense one-hot matrix followed by a matmul (`t_en @ pred_aln_trg`); the result is bit-identical because the
    matmul only ever adds zeros;
  * optional `durations=` forces the per-phoneme durations (throughput runs use 4 frames / phoneme so that
    every utterance is exactly 10 s) and removes the only data-dependent host sync.
"""
import torch

from . import _hooks, ops


def _engine_path(dev, taps=None):
    """The product path: every stage one C-ABI call into its C++ launch plan (csrc/st2_engine.hip).  The per-kernel Python
    plans run only where a plan has to be stepped through: tap points (`taps`), the tests' `_hooks.override(plan="python")`
    and the CPU plan tests (host tensors; the real kernel wrappers raise on those)."""
    return dev.type == "cuda" and taps is None and _hooks.plan == "engine"


class PartitionedStreams:
    """A pair of HIP streams on complementary compute-unit masks (st2_stream_create_cu_mask, include/st2.h): `front`
    owns `front_cus` of the device's CUs (dealt out evenly over the 8 XCDs by the driver's bit numbering), `main` the
    rest.  The two-stage pipeline of `inference(front_stream=...)` then does not depend on the hardware scheduler
    interleaving two queues.  MEASURED (round 3, DESIGN.md section 3 iv): slower than scheduler-placed two-stream execution
    on every box seen -- 126 / 90 / 75 ms per bench step with 16 / 32 / 64 front CUs against 68 (and 78 single-stream): the
    front's kernels are latency-bound but wide, a small partition starves them, a large one starves the decoder.  Kept as an
    explicit option (`bench.py --schedule partitioned`) for boxes on which two queues do not overlap.  Use as

        ps = PartitionedStreams(dev, front_cus=32)
        with torch.cuda.stream(ps.main):
            wave = inference(..., front_stream=ps.front)
    """

    def __init__(self, dev, front_cus, total_cus=None):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        dev = torch.device(dev)
        if total_cus is None:
            total_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if not 0 < front_cus < total_cus:
            raise ValueError("front_cus must be in (0, %d)" % total_cus)
        words = (total_cus + 31) // 32
        self.front_cus, self.total_cus = front_cus, total_cus
        self._handles = []
        with torch.cuda.device(dev):
            streams = []
            for lo, hi in ((0, front_cus), (front_cus, total_cus)):
                mask = (C.c_uint32 * words)()
                for b in range(lo, hi):
                    mask[b // 32] |= 1 << (b % 32)
                h = C.c_void_p()
                _lib.check(lib.st2_stream_create_cu_mask(mask, words, C.byref(h)), "st2_stream_create_cu_mask")
                self._handles.append(h)
                streams.append(torch.cuda.ExternalStream(h.value, device=dev))
        self.front, self.main = streams

    def close(self):
        from . import _lib
        lib = _lib.load()
        for h in self._handles:
            lib.st2_stream_destroy(h)
        self._handles = []


class MaskedStreams:
    """HIP streams confined to one CU mask (st2_stream_create_cu_mask): `main` and `front`, both on the SAME set of CUs --
    e.g. the device without the CUs `ops.probe_cu_health()` reports as slow.  Measured: in batch
    throughput a masked queue loses (the dispatcher still deals a masked XCD its eighth of every grid); no box has reported
    a slow CU since round 4's epilogue fix.  Use as

        rep, mask, n = ops.probe_cu_health()
        if n:
            ms = MaskedStreams(dev, mask)
            with torch.cuda.stream(ms.main):
                wave = inference(..., front_stream=ms.front)
    """

    def __init__(self, dev, mask_words, n_streams=2):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        dev = torch.device(dev)
        self._handles, streams = [], []
        with torch.cuda.device(dev):
            for _ in range(n_streams):
                mask = (C.c_uint32 * len(mask_words))(*mask_words)
                h = C.c_void_p()
                _lib.check(lib.st2_stream_create_cu_mask(mask, len(mask_words), C.byref(h)), "st2_stream_create_cu_mask")
                self._handles.append(h)
                streams.append(torch.cuda.ExternalStream(h.value, device=dev))
        self.main, self.front = streams[0], streams[1] if n_streams > 1 else None
        self.cus = sum(bin(w).count("1") for w in mask_words)

    def close(self):
        from . import _lib
        lib = _lib.load()
        for h in self._handles:
            lib.st2_stream_destroy(h)
        self._handles = []


def _pad_mask(lengths, N):
    """utils.length_to_mask (reference utils.py:42-46: True where position >= length) at a fixed width N: a bucketed
    batch may be wider than its longest utterance."""
    return torch.arange(N).unsqueeze(0) >= lengths.reshape(-1, 1)


def expand_by_durations(x, dur, T, shift=False):
    """x [B, C, N], dur [B, N] (int64, every row sums to T) -> [B, C, T] with frame t taking the phoneme whose frames
    cover it (== x @ one_hot alignment, Demo/Inference_LJSpeech.ipynb:303-312): one `st2_expand_by_durations` launch
    (prefix sum + binary search + gather on the device) instead of the notebook's Python loop and dense matmul.
    `shift`: the HiFi-GAN flow's one-frame right shift (Demo/Inference_LibriTTS.ipynb:306-319)."""
    return ops.expand_by_durations(x, dur.contiguous(), T, shift=shift)


def predict_durations(model, d, lj_tail=False, input_lengths=None):
    """ipynb:296-301: duration LSTM -> projection -> sum of sigmoids -> round, clamp(min=1), as two launches: the
    BiLSTM (k=1 conv input projection + recurrence) and `st2_duration_head` (Linear 512->50, sigmoid sum, round,
    clamp, pad masking and the LJSpeech +5 tail in one kernel).

    `input_lengths` (host int64 [B]) for a right-padded batch: the BiLSTM runs with packed-sequence semantics (its
    reverse direction starts at each utterance's own last token), pad positions get duration 0 and the LJSpeech
    +5-frame tail lands on each utterance's own last token -- every row is then what the notebook computes for that
    utterance alone."""
    B, N = d.shape[0], d.shape[1]
    if input_lengths is not None and input_lengths.is_cuda:  # device copy of a batch known to be padded
        lens = input_lengths.to(torch.int32)
    else:
        ragged = input_lengths is not None and not bool((input_lengths == N).all())
        lens = input_lengths.to(torch.int32).to(d.device) if ragged else None
    x = model.predictor.lstm.forward_cm(d.transpose(1, 2).contiguous().float(), lens)   # [B, 512, N]
    lin = model.predictor.duration_proj.linear_layer
    return ops.duration_head(x, lin.weight.detach().float().contiguous(), lin.bias.detach().float().contiguous(),
                             lengths=lens, tail=5 if lj_tail else 0)  # LJSpeech notebook only: pred_dur[-1] += 5


@torch.no_grad()
def _front_engine(model, dev):
    """The st2_engine handle of the front (text encoder, PL-BERT + bert_encoder, style denoiser, prosody predictor), packed
    once per (weights, device): rebuilt when a front module's parameters were reloaded (in-place version counters) or
    moved (storage addresses)."""
    from . import engine
    mods = [model.text_encoder, model.bert, model.bert_encoder, model.diffusion.diffusion.net, model.predictor]
    stamp = tuple((p.data_ptr(), p._version) for m in mods for p in m.parameters())
    dev = engine.norm_device(dev)
    cached = getattr(model.predictor, "_front_engine", None)
    if cached is None or cached[0] != stamp or not engine.same_device(cached[1], dev):
        engine.replaced(cached[1] if cached else None, "front")
        cached = (stamp, engine.build_front_engine(model, dev))
        model.predictor._front_engine = cached
    return cached[1]


def _front_core(model, sampler, tokens, lengths_host, lengths_dev, noise, step_noise, ref_s, s_prev, *, diffusion_steps,
                embedding_scale, alpha, beta, t, predict, lj_tail, carry=False, taps=None):
    """The device-only part of the front: text encoder, PL-BERT, style diffusion, style mixing, duration encoder and
    (when `predict`) the duration head.  No host read of device data, no host -> device copy, no random draw: every
    input is a device tensor (`lengths_dev` int32 [B] for a right-padded batch, else None and `lengths_host` decides on
    the host), so the whole function is legal under stream capture (`GraphedFront`).

    `carry`: the B rows are consecutive sentences of ONE passage; row k's sampled style is mixed with row k-1's mixed style
    (`s_prev` [1, 256] or None feeds row 0) -- LFinference's loop as a row scan between the batched sampler and the batched
    duration stages (st2.h st2_front_args.carry)."""
    dev = tokens.device
    B, N = tokens.shape
    if _engine_path(dev, taps):  # ONE C-ABI call: st2_front_forward (csrc/st2_engine.hip front_plan)
        from .diffusion import GraphedSampler
        smp = sampler.sampler if isinstance(sampler, GraphedSampler) else sampler
        if lengths_dev is None and lengths_host is not None and not bool((lengths_host == N).all()):
            lengths_dev = lengths_host.to(torch.int32).to(dev)
        if step_noise is None:
            step_noise = torch.randn((diffusion_steps - 1, B, 1, noise.shape[-1]), device=dev, dtype=torch.float32)
        table, sigma0 = smp.step_table(diffusion_steps)
        o = _front_engine(model, dev).front_forward(tokens, noise, step_noise, table, sigma0, lengths=lengths_dev, ref_s=ref_s,
                                                    s_prev=s_prev, embedding_scale=embedding_scale, alpha=alpha, beta=beta,
                                                    t=t, predict=predict, tail=5 if lj_tail else 0, carry=carry)
        return dict(t_en=o["t_en"], d=o["d_cm"].transpose(1, 2), s=o["s"], ref=o["ref"], durations=o["durations"],
                    s_mixed=o["s_pred"])  # [B, 2 sty] = (ref | s), written by the plan: no torch.cat on the product path
    if lengths_dev is not None:  # mask built on the device; the modules take the device copy (text.py _device_lengths)
        text_mask = torch.arange(N, device=dev).unsqueeze(0) >= lengths_dev.reshape(-1, 1)
        len_arg = lengths_dev
    else:
        text_mask = torch.zeros((B, N), dtype=torch.bool, device=dev)
        len_arg = lengths_host
    t_en = model.text_encoder(tokens, len_arg, text_mask)                            # [B, 512, N]
    bert_dur = model.bert(tokens, attention_mask=(~text_mask).int())                 # [B, N, 768]
    d_en = model.bert_encoder(bert_dur).transpose(-1, -2)                            # [B, 512, N]
    kw = dict(embedding=bert_dur, embedding_scale=embedding_scale, num_steps=diffusion_steps, step_noise=step_noise)
    if ref_s is not None:
        kw["features"] = ref_s
    if lengths_dev is not None:  # the denoiser attends over / averages each utterance's own tokens only
        kw["lengths"] = lengths_dev
    s_pred = sampler(noise, **kw).squeeze(1)                                          # [B, 256]
    if taps is not None:
        taps["s_pred"] = s_pred
    def mix(sp, prev, rs):
        if prev is not None:
            sp = t * prev + (1 - t) * sp  # convex combination of previous and current style
        s, ref = sp[:, 128:], sp[:, :128]
        if rs is not None:
            ref = alpha * ref + (1 - alpha) * rs[:, :128]
            s = beta * s + (1 - beta) * rs[:, 128:]
        return s, ref

    if carry and B > 1:  # row scan: sentence k mixes with sentence k-1's MIXED style
        rows, prev = [], s_prev
        for k in range(B):
            s_k, ref_k = mix(s_pred[k:k + 1], prev, None if ref_s is None else ref_s[k:k + 1])
            prev = torch.cat([ref_k, s_k], dim=-1)
            rows.append(prev)
        mixed = torch.cat(rows, dim=0)
        s, ref = mixed[:, 128:], mixed[:, :128]
    else:
        s, ref = mix(s_pred, s_prev, ref_s)
    s, ref = s.contiguous(), ref.contiguous()
    d = model.predictor.text_encoder(d_en, s, len_arg, text_mask)                    # [B, N, 640]
    dur = predict_durations(model, d, lj_tail=lj_tail, input_lengths=len_arg) if predict else None
    return dict(t_en=t_en, d=d, s=s, ref=ref, durations=dur)


class GraphedFront:
    """hipGraph replay of `_front_core` (BASELINE.json configs[4]: latency-bound sentence-by-sentence synthesis).  The
    front of one sentence is ~600 launches of 5-40 us kernels issued from Python (~6 ms of host time against ~4 ms of
    device time at B = 1): the host, not the GPU, paces a passage.  One graph per signature (batch, padded tokens,
    diffusion steps, guidance scale, speaker / carry-over / padding / duration-prediction flags, mixing weights); the
    first call with a new signature runs eagerly once and records, later calls copy their inputs into static buffers,
    replay and return CLONES of the outputs (the next replay may start while the decoder still reads them).  The
    per-step diffusion noise is always an explicit input (drawn here when the caller gives none)."""

    def __init__(self, model, sampler, max_graphs=32):
        self.model = model
        from .diffusion import GraphedSampler
        # a GraphedSampler's eager sampler: one graph, not two nested (DiffusionSampler.sampler is the ADPM2 object)
        self.sampler = sampler.sampler if isinstance(sampler, GraphedSampler) else sampler
        self.max_graphs = max_graphs
        self._graphs = {}

    def _generation(self):
        return getattr(self.sampler.diffusion.net, "_pack_gen", 0)

    def _pack_refs(self):
        """(module, packed-weight cache) of every front module as of now.  A graph keeps these references -- the device
        memory its kernels read stays allocated -- and is stale as soon as a module holds a different cache object
        (load_state_dict / .to() / refresh() rebuild the packs)."""
        refs = []
        for key in ("text_encoder", "bert", "predictor"):
            for m in self.model[key].modules():
                pk = getattr(m, "_pk", None)
                if pk is not None:
                    refs.append((m, pk))
        return refs

    def _engine_mode(self):
        return _hooks.plan == "engine"

    def _stale(self, g, dev):
        if g["engine"] is not None or self._engine_mode():  # recorded over / now running on the C++ front: same handle?
            if not self._engine_mode():
                return True
            eng = _front_engine(self.model, dev)
            return eng is not g["engine"] or eng.calib_gen != g["calib_gen"]  # the operand scales are kernel arguments
        return any(getattr(m, "_pk", None) is not pk for m, pk in g["packs"])

    @torch.no_grad()
    def __call__(self, tokens, lengths_host, lengths_dev, noise, step_noise, ref_s, s_prev, **kw):
        dev = tokens.device
        B, N = tokens.shape
        steps = kw["diffusion_steps"]
        if step_noise is None:
            step_noise = torch.randn((steps - 1, B, 1, noise.shape[-1]), device=dev, dtype=torch.float32)
        key = (dev.index, B, N, steps, float(kw["embedding_scale"]), ref_s is not None, s_prev is not None,
               lengths_dev is not None, bool(kw["predict"]), bool(kw["lj_tail"]), float(kw["alpha"]), float(kw["beta"]),
               float(kw["t"]), bool(kw.get("carry", False)))
        g = self._graphs.get(key)
        if g is not None and (g["gen"] != self._generation() or self._stale(g, dev)):  # packed weights were rebuilt
            self._graphs.clear()
            g = None
        if g is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            g = self._graphs[key] = self._capture(tokens, lengths_host, lengths_dev, noise, step_noise, ref_s, s_prev, kw)
        st = g["static"]
        for name, val in (("tokens", tokens), ("lengths_dev", lengths_dev), ("noise", noise), ("step_noise", step_noise),
                          ("ref_s", ref_s), ("s_prev", s_prev)):
            if val is not None:
                st[name].copy_(val.reshape(st[name].shape))
        g["graph"].replay()
        return {k: (None if v is None else v.clone()) for k, v in g["out"].items()}

    def _capture(self, tokens, lengths_host, lengths_dev, noise, step_noise, ref_s, s_prev, kw):
        clone = lambda v: None if v is None else v.detach().clone()
        st = dict(tokens=clone(tokens), lengths_dev=clone(lengths_dev), noise=clone(noise.float()),
                  step_noise=clone(step_noise.float()), ref_s=clone(ref_s), s_prev=clone(s_prev))
        lh = None if lengths_host is None else lengths_host.clone()

        def run():
            return _front_core(self.model, self.sampler, st["tokens"], lh, st["lengths_dev"], st["noise"],
                               st["step_noise"], st["ref_s"], st["s_prev"], **kw)

        dev = tokens.device
        cur = torch.cuda.current_stream(dev)
        side = ops.aux_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            run()  # eager warm-up on a side stream: weight packing, kernel attributes, allocator warm-up
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = run()
        # the graph's kernels read the packed weights: keep the Python caches / the C++ engine handle they live in alive,
        # and compare identities at replay (a reload or .to() rebuilds them)
        eng = _front_engine(self.model, dev) if self._engine_mode() else None
        return dict(graph=graph, static=st, out=out, gen=self._generation(), packs=self._pack_refs(), engine=eng,
                    calib_gen=None if eng is None else eng.calib_gen)


@torch.no_grad()
def prepare(model, sampler, tokens, input_lengths=None, noise=None, diffusion_steps=5, embedding_scale=1.0,
            ref_s=None, alpha=0.3, beta=0.7, durations=None, step_noise=None, lj_tail=None, s_prev=None, t=0.7,
            taps=None, allow_ragged=False, total_frames=None, lengths_dev=None, front=None, carry=False, group_events=False):
    """Everything in front of the decoder: text encoder, PL-BERT, style diffusion, style mixing, duration and
    prosody prediction, alignment expansion.  Returns the decoder's inputs {asr, F0, N, ref} plus the mixed style
    vector `s_pred` [B, 256] (what LFinference hands to the next sentence) and the durations.

    `s_prev` / `t`: long-form style carry-over, `s_pred = t * s_prev + (1 - t) * s_pred` applied to the sampler output
    before the speaker mixing (Demo/Inference_LibriTTS.ipynb LFinference; the LJSpeech notebook calls the same weight
    `alpha`).

    Utterances of different total duration cannot share a decoder call (its InstanceNorm spans the utterance).  With
    `allow_ragged` the result then carries `groups`: a list of (utterance indices, {asr, F0, N, ref}) per distinct
    frame count; without it such a batch raises.

    Host <-> device traffic: none on a batch without padding and with forced `durations` that are either a host tensor
    or a device tensor accompanied by `total_frames` (int, or one int per utterance: their row sums).  A pageable
    host -> device copy blocks the host until the stream has drained, i.e. it would serialise the issue of step k+1 with
    the execution of step k; so the pad mask of an unpadded batch is created on the device, and a caller that
    synthesises batch after batch keeps its forced durations on the device.

    `front` (a `GraphedFront`): the device-only part (`_front_core`) is replayed from a hipGraph instead of being issued
    kernel by kernel.

    `carry`: the rows are consecutive sentences of one passage (`_front_core`); `s_prev` is then [1, 256] or None.
    `group_events`: every entry of `groups` also carries `ready`, an event recorded on the current stream behind that group's
    last kernel (a consumer on another stream starts on group 0 while the later groups are still being expanded)."""
    dev = tokens.device
    B, N = tokens.shape
    ops.check_status() if dev.type == "cuda" else None  # device-side conditions raised by the previous call's kernels
    if input_lengths is None:
        input_lengths = torch.full((B,), N, dtype=torch.long)
    input_lengths = input_lengths.detach().cpu().long()
    ragged_n = not bool((input_lengths == N).all())
    if ragged_n and lengths_dev is None:  # ONE host -> device copy of the lengths (none if the caller prepared it)
        lengths_dev = input_lengths.to(torch.int32).to(dev)
    if not ragged_n:
        lengths_dev = None
    multispeaker = ref_s is not None
    hifigan = model.decoder.kind == "hifigan"
    if lj_tail is None:
        lj_tail = not multispeaker
    if noise is None:
        noise = torch.randn(B, 1, 256, device=dev)
    ckw = dict(diffusion_steps=diffusion_steps, embedding_scale=embedding_scale, alpha=alpha, beta=beta, t=t,
               predict=durations is None, lj_tail=lj_tail, carry=bool(carry))
    if front is not None and dev.type == "cuda" and taps is None:
        f = front(tokens, input_lengths, lengths_dev, noise, step_noise, ref_s, s_prev, **ckw)
    else:
        f = _front_core(model, sampler, tokens, input_lengths, lengths_dev, noise, step_noise, ref_s, s_prev, taps=taps,
                        **ckw)
    t_en, d, s, ref = f["t_en"], f["d"], f["s"], f["ref"]
    if durations is None:
        durations = f["durations"]
        tot = durations.sum(dim=1).tolist()  # the path's one data-dependent host sync: the frame counts
        ops.check_status() if dev.type == "cuda" else None  # everything up to here has completed: free to look
    else:
        durations = durations.long()
        if total_frames is not None:  # the caller knows the row sums: nothing is read back
            tot = [int(total_frames)] * B if isinstance(total_frames, int) else [int(v) for v in total_frames]
            assert len(tot) == B
        else:
            tot = durations.sum(dim=1).tolist()  # host tensor: no device sync; device tensor: one read-back (forced
            #                                      durations are the caller's: pad-token frames are expanded like any other)
    durations = durations.to(dev)
    if taps is not None:
        taps["durations"] = durations
    s_mixed = f.get("s_mixed")
    out = dict(ref=ref, s_pred=s_mixed if s_mixed is not None else torch.cat([ref, s], dim=-1), durations=durations)
    d_cm = d.transpose(-1, -2).contiguous()

    def expand(idx):
        """Alignment expansion + prosody for the utterances `idx` (all of one frame count T)."""
        T = int(tot[idx[0]])
        if len(idx) == B:
            sel = lambda v: v
        elif idx == list(range(idx[0], idx[0] + len(idx))):  # consecutive utterances: a view, no index tensor (whose pageable
            sel = lambda v: v[idx[0]:idx[0] + len(idx)]      # host -> device copy would stall the host behind the stream)
        else:
            sel = lambda v: v[torch.as_tensor(idx, device=dev)]
        dur = sel(durations)
        # hifigan: one-frame right shift, Demo/Inference_LibriTTS.ipynb:306-319
        if _engine_path(dev, taps):  # alignment expansion + F0Ntrain as ONE C-ABI call (st2_prosody_forward)
            asr, F0_pred, N_pred = _front_engine(model, dev).prosody_forward(sel(d_cm), sel(t_en), dur, sel(s), T,
                                                                             shift=hifigan)
            return dict(asr=asr, F0=F0_pred, N=N_pred, ref=sel(ref), en=None)
        en = expand_by_durations(sel(d_cm), dur, T, shift=hifigan)                    # [b, 640, T]
        asr = expand_by_durations(sel(t_en), dur, T, shift=hifigan)                   # [b, 512, T]
        F0_pred, N_pred = model.predictor.F0Ntrain(en, sel(s))
        return dict(asr=asr, F0=F0_pred, N=N_pred, ref=sel(ref), en=en)

    if len(set(tot)) == 1 and not group_events:
        g = expand(list(range(B)))
        if taps is not None:
            taps.update(F0=g["F0"], N=g["N"], asr=g["asr"], en=g["en"])
        out.update(asr=g["asr"], F0=g["F0"], N=g["N"])
        return out
    if not allow_ragged:
        raise ValueError("utterances of one call must have equal total duration; bucket them, pass `durations`, or "
                         "use inference() which decodes per frame count (frames per utterance: %s)" % tot)
    groups = {}
    for b, T in enumerate(tot):
        groups.setdefault(int(T), []).append(b)
    out["groups"] = []
    for idx in sorted(groups.values(), key=lambda v: v[0]):  # in the order of each group's first utterance
        g = expand(idx)
        if group_events and dev.type == "cuda":
            g["ready"] = torch.cuda.Event()
            g["ready"].record(torch.cuda.current_stream(dev))
        out["groups"].append((idx, g))
    return out


@torch.no_grad()
def inference(model, sampler, tokens, input_lengths=None, noise=None, diffusion_steps=5, embedding_scale=1.0,
              ref_s=None, alpha=0.3, beta=0.7, durations=None, step_noise=None, sine_noise=None, lj_tail=None,
              taps=None, front_stream=None, inputs_on_main=False, total_frames=None, front=None, decode_streams=None):
    """tokens [B, N] int64 (id 0 prepended, ipynb:277) -> waveform [B, 1, 600*T] on the device.

    Single-speaker (LJSpeech) when `ref_s` is None, else the multi-speaker flow with style mixing
    (Demo/Inference_LibriTTS.ipynb:285-286).  The decoder's InstanceNorm spans the whole utterance, so padding frames
    would change results (section 7.3-6): utterances whose (predicted) durations sum to different frame counts are
    decoded in one decoder call per distinct frame count and the result is then a LIST of B waveforms [1, 600*T_b] in
    utterance order; a batch of equal frame counts (forced `durations`, throughput runs) returns one tensor.  A
    right-padded batch (`input_lengths`) gives every utterance the result of its own un-padded run.

    `front_stream` (a torch.cuda.Stream): everything in front of the decoder is issued on that stream and handed to
    the decoder (on the current stream) through an event.  A caller that synthesises batch after batch thereby
    overlaps batch k+1's front -- a long chain of small, latency-bound kernels (BiLSTM recurrences on 64 CUs, 100-token
    transformer layers) -- with batch k's decoder, whose big convolutions fill whatever CUs the front leaves idle.
    Results are identical to the single-stream call.  The caller's input tensors must be complete when the front
    stream reaches them: inputs resident from earlier synchronised work (the bench, a server's staging buffers) need
    nothing; inputs still being produced on the CURRENT stream (a per-call torch.randn, an async H2D copy) need
    `inputs_on_main=True`, which makes the front stream wait for the current stream first -- at the price of also
    waiting for the previous call's decoder queued there, i.e. of the overlap.

    `decode_streams` (a list of torch streams; ragged batches only): the per-frame-count decoder calls -- independent of each
    other, one utterance each for real text -- are dealt round-robin onto these streams instead of running one after the other
    on the current one; the current stream waits for all of them before the call returns.  One utterance's decoder launches
    grids of 2 x 23 tiles for 256 CUs: two or three of them fill each other's idle CUs (the long-form path does the same,
    `synthesize_long(decode_streams=)`).  Bitwise the sequential result.
    """
    kw = dict(input_lengths=input_lengths, noise=noise, diffusion_steps=diffusion_steps,
              embedding_scale=embedding_scale, ref_s=ref_s, alpha=alpha, beta=beta, durations=durations,
              step_noise=step_noise, lj_tail=lj_tail, taps=taps, allow_ragged=True, total_frames=total_frames,
              front=front)
    if front_stream is None:
        p = prepare(model, sampler, tokens, **kw)
    else:
        main = torch.cuda.current_stream(tokens.device)
        if inputs_on_main:
            front_stream.wait_stream(main)
        with torch.cuda.stream(front_stream):
            p = prepare(model, sampler, tokens, **kw)
            ready = torch.cuda.Event()
            ready.record(front_stream)
        main.wait_event(ready)
        for g in ([p] if "groups" not in p else [g for _, g in p["groups"]]):
            for v in (g["asr"], g["F0"], g["N"], g["ref"]):
                v.record_stream(main)  # allocated on the front stream, consumed on the main stream
    if "groups" not in p:
        return model.decoder(p["asr"], p["F0"], p["N"], p["ref"], noise=sine_noise)
    waves = [None] * tokens.shape[0]
    dec = list(decode_streams) if decode_streams else []
    main = torch.cuda.current_stream(tokens.device) if dec else None
    for st in dec:
        st.wait_stream(main)  # the front's outputs (and the caller's inputs) are ordered on the current stream
    done = []
    for n_dec, (idx, g) in enumerate(p["groups"]):
        def run(idx=idx, g=g):
            sn = None if sine_noise is None else [sine_noise[b] for b in idx]
            if sn is not None:
                sn = torch.stack([n[:g["F0"].shape[1] * 300] for n in sn])
            return model.decoder(g["asr"], g["F0"], g["N"], g["ref"], noise=sn)
        if dec:
            ds = dec[n_dec % len(dec)]
            for v in (g["asr"], g["F0"], g["N"], g["ref"]):
                v.record_stream(ds)  # allocated on the current / front stream, consumed on the decoder's
            with torch.cuda.stream(ds):
                w = run()
                ev = torch.cuda.Event()
                ev.record(ds)
            w.record_stream(main)
            done.append(ev)
        else:
            w = run()
        for j, b in enumerate(idx):
            waves[b] = w[j]
    for ev in done:
        main.wait_event(ev)
    return waves


@torch.no_grad()
def synthesize_long(model, sampler, sentences, ref_s=None, alpha=0.3, beta=0.7, t=0.7, diffusion_steps=5,
                    embedding_scale=1.0, noises=None, step_noises=None, sine_noises=None, durations=None, trim=None,
                    overlap=True, on_chunk=None, bucket=0, front=None, side_stream=None, front_batch=1, decode_streams=1):
    """Long-form synthesis (BASELINE.json configs[4]; Demo/Inference_LibriTTS.ipynb LFinference + its driver loop,
    Demo/Inference_LJSpeech.ipynb "Long-form generation"): `sentences` is a list of token tensors [N_i] (id 0
    prepended); each sentence is synthesised with the previous sentence's mixed style carried over
    (`s_pred = t * s_prev + (1 - t) * s_pred`) and its waveform is handed out as soon as it is ready.

    The passage is sequential in the style vector only, and that vector is final before the sentence's decoder
    runs.  So the engine streams in two stages on two HIP streams: the front of sentence k+1 (text encoder, PL-BERT,
    diffusion, duration / prosody prediction; its one host sync is the predicted frame count) is issued on a side
    stream while the decoder + vocoder of sentence k (>= 2/3 of the sentence's time) still occupies the main stream;
    an event hands the decoder inputs over.  Returns (list of waveforms [600*T_i - trim], final style [1, 256]);
    `on_chunk(k, wave)` is called per sentence, in sentence order, for streaming consumers.  `trim` samples are dropped from
    every sentence's end as the notebooks do ("weird pulse at the end of the model": 100 multi-speaker, 0 single-speaker).

    `front_batch`: how many consecutive sentences share ONE front call.  1 = the notebooks' schedule, sentence by sentence
    (lowest time to the first waveform).  0 / None = the whole passage, n = groups of n, a sequence = those group sizes in turn
    with the last one repeating ((2, 0): the first two sentences, then the rest while their decoders run): the sentences' text encoder, PL-BERT,
    style diffusion and duration stages run as one right-padded batch (pad tokens masked everywhere: every row is the
    sentence's own un-padded result) with the style carry-over as a row scan in between (`_front_core(carry=True)`) -- a
    100-token sentence alone leaves its ~1 500 token GEMMs and BiLSTM steps latency-bound, ten of them fill the same
    launches.  The alignment expansion / prosody predictor and the decoder still run per sentence (their InstanceNorm
    spans the utterance), in sentence order, each waiting only for its own inputs.

    `decode_streams` > 1 (with `overlap`): the decoder calls are dealt round-robin onto that many auxiliary streams instead
    of the caller's.  One sentence's decoder is ~400 launches whose grids cover a fraction of the chip (2 x 23 tiles for 256
    CUs); with a batched front the next sentences' inputs are ready long before, so independent sentences' decoders fill each
    other's idle CUs.  The caller's stream waits for sentence k's decoder before `on_chunk(k, ...)` and for all of them before
    the call returns: the waveforms are ordered on the caller's stream exactly as before.  A list of torch streams is used as
    given: HIP multiplexes streams onto ~4 hardware queues in creation order, and two decoder streams that share a queue with
    each other or with the front's stream serialise (or worse: 113 ms against 65 on one stream) -- a serving process
    measures its candidates once at start-up, as bench.py does.

    `bucket` > 0: every sentence's token row is right-padded to a multiple of `bucket` (the pad tokens are masked
    everywhere: packed-sequence BiLSTMs, key-padded attention, length-aware mean -- results are those of the un-padded
    sentence), so that a `GraphedSampler` (models.make_sampler(graph=True)) replays ONE captured hipGraph per bucket
    instead of capturing one per sentence length.
    """
    dev = sentences[0].device
    multispeaker = ref_s is not None
    if trim is None:
        trim = 100 if multispeaker else 0
    use_streams = overlap and dev.type == "cuda"
    main = torch.cuda.current_stream(dev) if use_streams else None
    side = (side_stream if side_stream is not None else ops.aux_stream(dev)) if use_streams else None  # (a caller may
    #                                                  hand in a CU-masked side stream: pipeline.MaskedStreams)
    K = len(sentences)
    sizes = list(front_batch) if isinstance(front_batch, (list, tuple)) else [front_batch]
    starts, i = [], 0
    while i < K:  # chunk sizes in turn, the last one repeating; 0 / None = everything that is left
        n = sizes[min(len(starts), len(sizes) - 1)]
        n = K - i if not n else max(1, min(int(n), K - i))
        starts.append((i, i + n))
        i += n
    # per-call inputs are prepared (padded, stacked, moved to the device) BEFORE the streaming loop: a pageable
    # host -> device copy inside it would block the host until the issuing stream has drained
    prepped = []
    for i, i_end in starts:
        ids = list(range(i, i_end))
        ns = [sentences[k].numel() for k in ids]
        npad = max(ns)
        if bucket and npad % bucket:
            npad = (npad + bucket - 1) // bucket * bucket
        tokens = torch.zeros((len(ids), npad), dtype=sentences[i].dtype, device=dev)  # token id 0 = pad (text_utils / ipynb:277)
        for j, k in enumerate(ids):
            tokens[j, :ns[j]] = sentences[k].reshape(-1)
        ragged = any(n != npad for n in ns)
        lengths = torch.LongTensor(ns) if ragged else None
        dur, frames = None, None
        if durations is not None:
            rows = [durations[k].reshape(-1).long() for k in ids]
            if not rows[0].is_cuda:
                frames = [int(r.sum()) for r in rows]
            dur = torch.zeros((len(ids), npad), dtype=torch.long, device=dev)  # pad tokens get no frames
            for j, r in enumerate(rows):
                dur[j, :ns[j]] = r.to(dev)
        cat = lambda seq, dim: None if seq is None else torch.cat([seq[k] for k in ids], dim=dim)
        prepped.append(dict(ids=ids, tokens=tokens, lengths=lengths, dur=dur, frames=frames,
                            lens_dev=None if lengths is None else lengths.to(torch.int32).to(dev),
                            noise=cat(noises, 0), step_noise=cat(step_noises, 1),
                            ref_s=None if ref_s is None else ref_s.reshape(1, -1).expand(len(ids), -1).contiguous()))
    dec = []
    if use_streams and isinstance(decode_streams, (list, tuple)):
        dec = list(decode_streams) if len(decode_streams) > 1 else []
    elif use_streams and decode_streams and int(decode_streams) > 1:
        dec = [ops.aux_stream(dev, 0, index=i + 1) for i in range(int(decode_streams))]
    if use_streams:
        # weights and inputs produced on the caller's stream -- including the rows stacked just above -- are visible to the other streams
        for st in [side] + dec:
            st.wait_stream(main)
        for q in prepped:
            for v in q.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(side)  # allocated on the caller's stream, read on the side stream
    s_prev, waves, emitted, n_dec, done = None, [None] * K, 0, 0, {}

    def emit(emitted):
        while emitted < K and waves[emitted] is not None:
            if done[emitted] is not None:
                main.wait_event(done[emitted])
            if on_chunk is not None:
                on_chunk(emitted, waves[emitted])
            emitted += 1
        return emitted
    for q in prepped:
        ids = q["ids"]
        kw = dict(input_lengths=q["lengths"], noise=q["noise"], diffusion_steps=diffusion_steps,
                  embedding_scale=embedding_scale, ref_s=q["ref_s"], alpha=alpha, beta=beta, lj_tail=False, s_prev=s_prev, t=t,
                  step_noise=q["step_noise"], durations=q["dur"], total_frames=q["frames"], lengths_dev=q["lens_dev"],
                  front=front, carry=len(ids) > 1, allow_ragged=True, group_events=use_streams and len(ids) > 1)
        if use_streams:
            with torch.cuda.stream(side):
                p = prepare(model, sampler, q["tokens"], **kw)
                ready = torch.cuda.Event()
                ready.record(side)
        else:
            p = prepare(model, sampler, q["tokens"], **kw)
        s_prev = p["s_pred"][-1:]
        groups = p["groups"] if "groups" in p else [(list(range(len(ids))), p)]
        for idx, g in groups:  # one decoder call per distinct frame count, in the order of each group's first sentence
            ds = dec[n_dec % len(dec)] if dec else main
            n_dec += 1
            if use_streams:
                ds.wait_event(g.get("ready", ready))
                for v in (g["asr"], g["F0"], g["N"], g["ref"]):
                    v.record_stream(ds)  # allocated on the side stream, consumed on the decoder's stream
            stack = lambda: None if sine_noises is None else torch.cat([sine_noises[ids[j]] for j in idx], dim=0)
            if dec:
                with torch.cuda.stream(ds):  # (the noise rows are stacked on the stream that reads them)
                    w = model.decoder(g["asr"], g["F0"], g["N"], g["ref"], noise=stack())
                    ev = torch.cuda.Event()
                    ev.record(ds)
                w.record_stream(main)  # handed to the caller's stream behind `ev`
            else:
                w, ev = model.decoder(g["asr"], g["F0"], g["N"], g["ref"], noise=stack()), None
            for j, b in enumerate(idx):
                wave = w[j].reshape(-1)
                waves[ids[b]] = wave[:-trim] if trim else wave
                done[ids[b]] = ev
            if not dec:
                emitted = emit(emitted)
        # Decoders on auxiliary streams: the caller's stream takes its per-sentence waits only AFTER every decoder of this front
        # group has been queued.  A wait packet sits at the head of the caller's hardware queue until its decoder is done, and HIP
        # multiplexes streams onto ~4 such queues: an auxiliary stream that shares the caller's queue had its NEXT decoder queued
        # behind that wait -- the decoders serialised, and a passage took 74-114 ms instead of 52 depending on which streams the
        # process happened to get (rounds 5-6: "stream roulette").  Queued last, the waits block nothing.
        if dec:
            emitted = emit(emitted)
    if use_streams and s_prev is not None:
        s_prev.record_stream(main)  # allocated on the side stream, handed to the caller's
    return waves, s_prev


# ---------------------------------------------------------------------------------------------------------------------
# per-layer operand scales of the split-f16 convs (include/st2.h st2_calibrate)
# ---------------------------------------------------------------------------------------------------------------------
def calibrate(run, margin_bits=3, max_passes=3, engines=None, accumulate=False):
    """Start-up calibration of a serving process: `run()` issues one or more representative forwards through the product
    path (e.g. `lambda: inference(model, sampler, tokens, ...)`, plus `style.compute_style(model, wave)` for a zero-shot
    model); every split-f16 conv of every live engine that was launched gets its own power-of-two operand scale from the
    largest input it saw (the reference's convs are fp32 at every magnitude, Modules/istftnet.py:68-74; by rule the scale is
    8 / 1, which is fp32-class for O(1) tensors only).  A pass whose launches ran into the f16 clamp at their old scale only
    bounds those layers from below, so the recording is repeated (at most `max_passes` times).  Calibrate BEFORE recording
    hipGraphs (GraphedFront re-records by itself); results stay bitwise reproducible for a given table.  `run()` should cover the
    spread of the traffic (several utterances, several reference styles): a site keeps 8 x headroom over the largest operand seen
    after a normalising prologue and 32 x where its input is free-ranging (F0 in Hz, stage outputs, FFN intermediates); beyond that
    the clamp + ST2_STATUS_F16_RANGE stay as the net.  `accumulate=True` widens the existing table with what this call sees instead
    of starting over (more utterances later, same checkpoint).  Returns
    {"passes", "sites_set", "clamped_last_pass", "headroom": rows of the last pass (ops.headroom)}."""
    from . import _lib, engine
    rows, nset, clamped, passes = [], 0, 0, 0
    ops.check_status()  # whatever earlier calls raised is theirs: reported now, not swallowed by the passes below
    if not accumulate:
        for eng in (engines if engines is not None else engine.live_engines()):
            eng.set_calibration(None)  # st2_calibrate accumulates maxima: a fresh calibration starts from the rule
    for _ in range(max(1, int(max_passes))):
        with ops.headroom() as h:
            run()
        ops.check_status(ignore=_lib.STATUS_F16_RANGE)  # a clamp during calibration is what the pass is there to find
        rows, passes = h.rows, passes + 1
        nset = clamped = 0
        for eng in (engines if engines is not None else engine.live_engines()):
            n, c = eng.calibrate(margin_bits)
            nset, clamped = nset + n, clamped + c
        if clamped == 0:
            break
    return {"passes": passes, "sites_set": nset, "clamped_last_pass": clamped, "headroom": rows}


def model_engines(model, dev):
    """{"front": ..., "decoder": ..., "style": ...}: the st2_engine handles behind a model's product path on `dev`, built if
    they are not yet (same caches as the forward calls use)."""
    from . import engine, style
    dev = engine.norm_device(dev)
    dec = model.decoder
    if getattr(dec, "_eng", None) is None or not engine.same_device(dec._eng, dev):
        engine.replaced(getattr(dec, "_eng", None), "decoder")
        dec._eng = engine.build_decoder_engine(dec, dev)
    out = {"front": _front_engine(model, dev), "decoder": dec._eng}
    se, pe = model.get("style_encoder"), model.get("predictor_encoder")
    on_dev = lambda m: all(engine.norm_device(p.device) == dev for p in m.parameters())
    if isinstance(se, style.StyleEncoder) and isinstance(pe, style.StyleEncoder) and on_dev(se) and on_dev(pe):
        out["style"] = style._style_engine(model, dev)  # (a process that never moved the style encoders to `dev` has no such engine)
    return out


def calibration_state(model, dev):
    """JSON-serialisable {"front": [x_scale per conv site], "decoder": [...], ...} (0.0 = by rule): what a process saves
    beside a checkpoint, and what rank 0 sends to the other ranks (`parallel.broadcast_calibration`)."""
    return {k: e.calibration_scales() for k, e in model_engines(model, dev).items()}


def load_calibration_state(model, dev, state):
    """Installs a table made by `calibration_state` on a process holding the same model (same conv layout: checked)."""
    engs = model_engines(model, dev)
    for k, scales in state.items():
        if k in engs:
            engs[k].set_calibration(scales)
