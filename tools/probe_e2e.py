"""GPU probe: per-stage wall time of the text->waveform pipeline at the bench workload (sync after each stage)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from _util import manifest
from styletts2_amd import models, pipeline
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
from styletts2_amd.utils import length_to_mask

tag = os.environ.get("PROBE_TAG", "ljspeech")
B = int(os.environ.get("PROBE_B", "32"))
N, steps = 100, int(os.environ.get("PROBE_STEPS", "5"))
dev = "cuda"
man = manifest(tag)
model = models.build_model(models.recursive_munch(man["config"]), None, None, models.load_plbert(man["plbert"]))
KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]
for i, k in enumerate(KEYS):
    synth.init_synthetic_(model[k], 10 + i)
    model[k].eval().to(dev)
sampler = models.make_sampler(model, graph=os.environ.get("PROBE_GRAPH", "0") == "1")
g = torch.Generator().manual_seed(0)
tokens = torch.randint(1, 178, (B, N), generator=g).to(dev)
lengths = torch.full((B,), N, dtype=torch.long)
noise = torch.randn(B, 1, 256, generator=g).to(dev)
dur = torch.full((B, N), 4, dtype=torch.long)
ref_s = torch.randn(B, 256, generator=g).to(dev) if man["config"]["multispeaker"] else None


def timed(name, fn, acc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r


for it in range(3):
    acc = {}
    mask = length_to_mask(lengths).to(dev)
    t_en = timed("text_encoder", lambda: model.text_encoder(tokens, lengths, mask), acc)
    bert_dur = timed("bert", lambda: model.bert(tokens, attention_mask=(~mask).int()), acc)
    d_en = timed("bert_encoder", lambda: model.bert_encoder(bert_dur).transpose(-1, -2), acc)
    kw = dict(embedding=bert_dur, embedding_scale=1.0, num_steps=steps)
    if ref_s is not None:
        kw["features"] = ref_s
    s_pred = timed("sampler", lambda: sampler(noise, **kw).squeeze(1), acc)
    s, ref = s_pred[:, 128:].contiguous(), s_pred[:, :128].contiguous()
    d = timed("duration_encoder", lambda: model.predictor.text_encoder(d_en, s, lengths, mask), acc)
    timed("duration_head", lambda: pipeline.predict_durations(model, d), acc)
    T = 4 * N
    en = timed("expand", lambda: pipeline.expand_by_durations(d.transpose(-1, -2).contiguous(), dur.to(dev), T), acc)
    asr = timed("expand", lambda: pipeline.expand_by_durations(t_en, dur.to(dev), T), acc)
    F0, Nn = timed("F0Ntrain", lambda: model.predictor.F0Ntrain(en, s), acc)
    out = timed("decoder", lambda: model.decoder(asr, F0, Nn, ref), acc)
    tot = sum(acc.values())
    print("iter %d total %.1f ms (%.0f audio-s/s): " % (it, tot, B * 10.0 / tot * 1e3) +
          ", ".join("%s %.1f" % kv for kv in acc.items()), flush=True)
    t0 = time.perf_counter()
    out = pipeline.inference(model, sampler, tokens, lengths, noise, diffusion_steps=steps, ref_s=ref_s, durations=dur)
    torch.cuda.synchronize()
    print("   unsynced pipeline.inference: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
