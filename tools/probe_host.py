"""GPU probe: where the HOST time of one text -> waveform step goes (bench configuration configs[1], B = 32).  Per stage:
wall time until the stage's Python call returns with the GPU idle at the start and nothing waited for (= issue time);
then a cProfile of five whole steps, top functions by cumulative time."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from _util import manifest
from styletts2_amd import models, pipeline
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)

dev = "cuda"
B = int(os.environ.get("PROBE_B", "32"))
man = manifest(os.environ.get("PROBE_TAG", "ljspeech"))
model = models.build_model(models.recursive_munch(man["config"]), None, None, models.load_plbert(man["plbert"]))
KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]
for i, k in enumerate(KEYS):
    synth.init_synthetic_(model[k], 10 + i)
    model[k].eval().to(dev)
sampler = models.make_sampler(model)
g = torch.Generator().manual_seed(0)
N = 100
tokens = torch.randint(1, 178, (B, N), generator=g)
tokens[:, 0] = 0
tokens = tokens.to(dev)
lengths = torch.full((B,), N, dtype=torch.long)
noise = torch.randn(B, 1, 256, generator=g).to(dev)
durations = torch.full((B, N), 4, dtype=torch.long)
ref_s = torch.randn(B, 256, generator=g).to(dev) if man["config"]["multispeaker"] else None


def step():
    return pipeline.inference(model, sampler, tokens, lengths, noise, diffusion_steps=5, ref_s=ref_s, durations=durations)


for _ in range(2):
    step()
torch.cuda.synchronize()

mask = pipeline._pad_mask(lengths, N).to(dev)
stages = {}


def stage(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    stages.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return r


for rep in range(4):
    t_en = stage("text_encoder", lambda: model.text_encoder(tokens, lengths, mask))
    bert_dur = stage("bert", lambda: model.bert(tokens, attention_mask=(~mask).int()))
    d_en = stage("bert_encoder", lambda: model.bert_encoder(bert_dur).transpose(-1, -2))
    kw = dict(embedding=bert_dur, embedding_scale=1.0, num_steps=5)
    if ref_s is not None:
        kw["features"] = ref_s
    s_pred = stage("sampler", lambda: sampler(noise, **kw).squeeze(1))
    s, ref = s_pred[:, 128:].contiguous(), s_pred[:, :128].contiguous()
    d = stage("duration_encoder", lambda: model.predictor.text_encoder(d_en, s, lengths, mask))
    stage("duration_head", lambda: pipeline.predict_durations(model, d, lj_tail=True, input_lengths=lengths))
    dur = durations.to(dev)
    hif = model.decoder.kind == "hifigan"
    en = stage("expand", lambda: pipeline.expand_by_durations(d.transpose(-1, -2).contiguous(), dur, 400, shift=hif))
    asr = pipeline.expand_by_durations(t_en, dur, 400, shift=hif)
    F0, Nn = stage("F0Ntrain", lambda: model.predictor.F0Ntrain(en, s))
    stage("decoder", lambda: model.decoder(asr, F0, Nn, ref))
torch.cuda.synchronize()
tot = 0.0
for k, v in stages.items():
    print("host issue  %-18s %7.3f ms (min of %d)" % (k, min(v), len(v)))
    tot += min(v)
print("host issue  %-18s %7.3f ms" % ("sum", tot))

torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
