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

from .utils import length_to_mask


def expand_by_durations(x, dur, T):
    """x [B, C, N], dur [B, N] (int64, every row sums to T) -> [B, C, T] with frame t taking phoneme idx[t]
    (== x @ one_hot alignment, Demo/Inference_LJSpeech.ipynb:303-312)."""
    B, C, N = x.shape
    ar = torch.arange(N, device=x.device)
    idx = torch.stack([torch.repeat_interleave(ar, dur[b], output_size=T) for b in range(B)])  # [B, T]
    return torch.gather(x, 2, idx.unsqueeze(1).expand(B, C, T))


def predict_durations(model, d, lj_tail=False):
    """ipynb:296-301: duration LSTM -> projection -> sum of sigmoids -> round, clamp(min=1)."""
    x, _ = model.predictor.lstm(d)
    duration = model.predictor.duration_proj(x)
    duration = torch.sigmoid(duration).sum(dim=-1)
    pred_dur = torch.round(duration).clamp(min=1).long()
    if lj_tail:
        pred_dur[:, -1] += 5  # LJSpeech notebook only (ipynb:301)
    return pred_dur


@torch.no_grad()
def prepare(model, sampler, tokens, input_lengths=None, noise=None, diffusion_steps=5, embedding_scale=1.0,
            ref_s=None, alpha=0.3, beta=0.7, durations=None, step_noise=None, lj_tail=None, s_prev=None, t=0.7,
            taps=None):
    """Everything in front of the decoder: text encoder, PL-BERT, style diffusion, style mixing, duration and
    prosody prediction, alignment expansion.  Returns the decoder's inputs {asr, F0, N, ref} plus the mixed style
    vector `s_pred` [B, 256] (what LFinference hands to the next sentence) and the durations.

    `s_prev` / `t`: long-form style carry-over, `s_pred = t * s_prev + (1 - t) * s_pred` applied to the sampler output
    before the speaker mixing (Demo/Inference_LibriTTS.ipynb LFinference; the LJSpeech notebook calls the same weight
    `alpha`)."""
    dev = tokens.device
    B, N = tokens.shape
    if input_lengths is None:
        input_lengths = torch.full((B,), N, dtype=torch.long)
    text_mask = length_to_mask(input_lengths).to(dev)
    multispeaker = ref_s is not None
    hifigan = model.decoder.kind == "hifigan"
    if lj_tail is None:
        lj_tail = not multispeaker
    if noise is None:
        noise = torch.randn(B, 1, 256, device=dev)

    t_en = model.text_encoder(tokens, input_lengths, text_mask)                      # [B, 512, N]
    bert_dur = model.bert(tokens, attention_mask=(~text_mask).int())                 # [B, N, 768]
    d_en = model.bert_encoder(bert_dur).transpose(-1, -2)                            # [B, 512, N]

    kw = dict(embedding=bert_dur, embedding_scale=embedding_scale, num_steps=diffusion_steps, step_noise=step_noise)
    if multispeaker:
        kw["features"] = ref_s
    s_pred = sampler(noise, **kw).squeeze(1)                                          # [B, 256]
    if taps is not None:
        taps["s_pred"] = s_pred
    if s_prev is not None:
        s_pred = t * s_prev + (1 - t) * s_pred  # convex combination of previous and current style
    s = s_pred[:, 128:]
    ref = s_pred[:, :128]
    if multispeaker:
        ref = alpha * ref + (1 - alpha) * ref_s[:, :128]
        s = beta * s + (1 - beta) * ref_s[:, 128:]
    s, ref = s.contiguous(), ref.contiguous()

    d = model.predictor.text_encoder(d_en, s, input_lengths, text_mask)              # [B, N, 640]
    if durations is None:
        durations = predict_durations(model, d, lj_tail=lj_tail)
        tot = durations.sum(dim=1)
        if not bool((tot == tot[0]).all()):
            raise ValueError("utterances of one call must have equal total duration; bucket them or pass "
                             "`durations` (frames per utterance: %s)" % tot.tolist())
    T = int(durations[0].sum())  # host value when `durations` was given on the host: no device sync
    durations = durations.to(dev)
    if taps is not None:
        taps["durations"] = durations

    en = expand_by_durations(d.transpose(-1, -2).contiguous(), durations, T)          # [B, 640, T]
    asr = expand_by_durations(t_en, durations, T)                                      # [B, 512, T]
    if hifigan:  # one-frame right shift, Demo/Inference_LibriTTS.ipynb:306-319
        en = torch.cat([en[:, :, :1], en[:, :, :-1]], dim=2)
        asr = torch.cat([asr[:, :, :1], asr[:, :, :-1]], dim=2)
    F0_pred, N_pred = model.predictor.F0Ntrain(en.contiguous(), s)
    if taps is not None:
        taps.update(F0=F0_pred, N=N_pred, asr=asr, en=en)
    return dict(asr=asr.contiguous(), F0=F0_pred, N=N_pred, ref=ref, s_pred=torch.cat([ref, s], dim=-1),
                durations=durations)


@torch.no_grad()
def inference(model, sampler, tokens, input_lengths=None, noise=None, diffusion_steps=5, embedding_scale=1.0,
              ref_s=None, alpha=0.3, beta=0.7, durations=None, step_noise=None, sine_noise=None, lj_tail=None,
              taps=None, front_stream=None):
    """tokens [B, N] int64 (id 0 prepended, ipynb:277) -> waveform [B, 1, 600*T] on the device.

    Single-speaker (LJSpeech) when `ref_s` is None, else the multi-speaker flow with style mixing
    (Demo/Inference_LibriTTS.ipynb:285-286).  All utterances of one call must expand to the same number of
    frames (the decoder's InstanceNorm spans the whole utterance, so padding would change results: section 7.3-6);
    callers bucket by length or pass `durations`.

    `front_stream` (a torch.cuda.Stream): everything in front of the decoder is issued on that stream and handed to
    the decoder (on the current stream) through an event.  A caller that synthesises batch after batch thereby
    overlaps batch k+1's front -- a long chain of small, latency-bound kernels (BiLSTM recurrences on 64 CUs, 100-token
    transformer layers) -- with batch k's decoder, whose big convolutions fill whatever CUs the front leaves idle.
    Results are identical to the single-stream call.
    """
    kw = dict(input_lengths=input_lengths, noise=noise, diffusion_steps=diffusion_steps,
              embedding_scale=embedding_scale, ref_s=ref_s, alpha=alpha, beta=beta, durations=durations,
              step_noise=step_noise, lj_tail=lj_tail, taps=taps)
    if front_stream is None:
        p = prepare(model, sampler, tokens, **kw)
    else:
        main = torch.cuda.current_stream(tokens.device)
        with torch.cuda.stream(front_stream):
            p = prepare(model, sampler, tokens, **kw)
            ready = torch.cuda.Event()
            ready.record(front_stream)
        main.wait_event(ready)
        for v in (p["asr"], p["F0"], p["N"], p["ref"]):
            v.record_stream(main)  # allocated on the front stream, consumed on the main stream
    return model.decoder(p["asr"], p["F0"], p["N"], p["ref"], noise=sine_noise)


@torch.no_grad()
def synthesize_long(model, sampler, sentences, ref_s=None, alpha=0.3, beta=0.7, t=0.7, diffusion_steps=5,
                    embedding_scale=1.0, noises=None, step_noises=None, sine_noises=None, durations=None, trim=None,
                    overlap=True, on_chunk=None):
    """Long-form synthesis (BASELINE.json configs[4]; Demo/Inference_LibriTTS.ipynb LFinference + its driver loop,
    Demo/Inference_LJSpeech.ipynb "Long-form generation"): `sentences` is a list of token tensors [N_i] (id 0
    prepended); each sentence is synthesised with the previous sentence's mixed style carried over
    (`s_pred = t * s_prev + (1 - t) * s_pred`) and its waveform is handed out as soon as it is ready.

    The passage is sequential in the style vector only, and that vector is final before the sentence's decoder
    runs.  So the engine streams in two stages on two HIP streams: the front of sentence k+1 (text encoder, PL-BERT,
    diffusion, duration / prosody prediction; its one host sync is the predicted frame count) is issued on a side
    stream while the decoder + vocoder of sentence k (>= 2/3 of the sentence's time) still occupies the main stream;
    an event hands the decoder inputs over.  Returns (list of waveforms [600*T_i - trim], final style [1, 256]);
    `on_chunk(k, wave)` is called per sentence for streaming consumers.  `trim` samples are dropped from every
    sentence's end as the notebooks do ("weird pulse at the end of the model": 100 multi-speaker, 0 single-speaker).
    """
    dev = sentences[0].device
    multispeaker = ref_s is not None
    if trim is None:
        trim = 100 if multispeaker else 0
    use_streams = overlap and dev.type == "cuda"
    main = torch.cuda.current_stream(dev) if use_streams else None
    side = torch.cuda.Stream(dev) if use_streams else None
    if use_streams:
        side.wait_stream(main)  # weights / inputs produced on the main stream are visible to the side stream
    s_prev, waves = None, []
    for k, tok in enumerate(sentences):
        tokens = tok.reshape(1, -1)
        noise = noises[k] if noises is not None else None
        kw = dict(noise=noise, diffusion_steps=diffusion_steps, embedding_scale=embedding_scale, ref_s=ref_s,
                  alpha=alpha, beta=beta, lj_tail=False, s_prev=s_prev, t=t,
                  step_noise=step_noises[k] if step_noises is not None else None,
                  durations=durations[k] if durations is not None else None)
        if use_streams:
            with torch.cuda.stream(side):
                p = prepare(model, sampler, tokens, **kw)
                ready = torch.cuda.Event()
                ready.record(side)
            main.wait_event(ready)
            for v in (p["asr"], p["F0"], p["N"], p["ref"]):
                v.record_stream(main)  # allocated on the side stream, consumed on the main stream
        else:
            p = prepare(model, sampler, tokens, **kw)
        s_prev = p["s_pred"]
        wave = model.decoder(p["asr"], p["F0"], p["N"], p["ref"],
                             noise=sine_noises[k] if sine_noises is not None else None)
        wave = wave.reshape(-1)
        if trim:
            wave = wave[:-trim]
        waves.append(wave)
        if on_chunk is not None:
            on_chunk(k, wave)
    return waves, s_prev
