#!/usr/bin/env python
"""End-to-end throughput bench: audio-seconds per wall-second (RTF^-1) of the text->waveform hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LJSpeech single-speaker model, iSTFTNet decoder, 5 diffusion steps, a batch of
32 synthetic 100-phoneme sequences per GPU with durations forced to 4 frames / phoneme (=> exactly 10.0 s of 24 kHz
audio per utterance), seeded random weights of the reference architecture (no checkpoints offline).  One "step" is
one full pass token ids -> waveform over the per-GPU batch; inputs are resident in HBM, outputs stay in HBM, the
random draws the reference makes inside forward (SineGen noise, ADPM2 step noise) are made inside the timed region.
Multi-GPU is weak scaling: every rank synthesises its own 32 utterances, no collective in steady state; the only
collective is the start-up weight broadcast over RCCL.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

PER_GPU_BATCH = 32
N_PHONEMES = 100
FRAMES_PER_PHONEME = 4
DIFFUSION_STEPS = 5
AUDIO_S_PER_UTT = N_PHONEMES * FRAMES_PER_PHONEME * 600 / 24000.0  # 10.0
# MI355X_MICROARCH.md: dense f16/bf16 MFMA peak ~2.5 PFLOP/s (v_mfma_f32_32x32x16_f16, 1024 FLOP/clk/SIMD).  The
# dominant kernel evaluates every fp32-class multiply as THREE f16 MFMA products (hi*hi + hi*lo + lo*hi, fp32
# accumulate), so the roof for ALGORITHMIC conv FLOPs is a third of that.
F16_MFMA_PEAK_TFLOPS = 2500.0
F16S_PRODUCTS = 3
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_dominant.json")  # written by tools/pmc_summary.py --json (rocprofv3 --pmc passes)
KEYS = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]


def build(man, seed_base=10):
    from styletts2_amd import models, synth
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    return model


def synthetic_inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(1, 178, (B, N_PHONEMES), generator=g)
    tokens[:, 0] = 0  # the notebooks prepend the pad id, Demo/Inference_LJSpeech.ipynb:277
    noise = torch.randn(B, 1, 256, generator=g)
    durations = torch.full((B, N_PHONEMES), FRAMES_PER_PHONEME, dtype=torch.long)
    lengths = torch.full((B,), N_PHONEMES, dtype=torch.long)
    return tokens, lengths, noise, durations


T0 = time.time()


def log(msg):
    """Progress to stderr (stdout carries exactly one JSON line)."""
    print("[bench %7.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


def cpu_baseline(man, sds):
    """The oracle (a CPU restatement of the reference path, kind="port") timed on the host cores on a bounded
    sample of the same workload: ONE 10 s utterance (BASELINE.json configs[0]), 1 warm-up + best of 3."""
    from oracle import st2_oracle as O
    threads = min(os.cpu_count() or 1, int(os.environ.get("ST2_CPU_THREADS", "64")))
    torch.set_num_threads(threads)  # oneDNN/MKL stop scaling (and start thrashing) far below 256 threads
    tokens, lengths, noise, durations = synthetic_inputs(1, 0)
    g = torch.Generator().manual_seed(1)
    step_noise = torch.randn(DIFFUSION_STEPS - 1, 1, 1, 256, generator=g)
    sine_noise = torch.randn(1, int(AUDIO_S_PER_UTT * 24000), 9, generator=g)
    best = None
    for it in range(4):
        t0 = time.time()
        with torch.no_grad():
            O.inference(sds, man["config"], man["plbert"], tokens, lengths, noise, step_noise, sine_noise,
                        diffusion_steps=DIFFUSION_STEPS, durations=durations)
        dt = time.time() - t0
        log("cpu_baseline run %d: %.2f s on %d threads" % (it, dt, threads))
        if it > 0 or dt > 30.0:
            best = dt if best is None else min(best, dt)
        if dt > 30.0:  # keep the bench bounded on a slow host: a single (cold) run is reported as such
            break
    return {"value": AUDIO_S_PER_UTT / best, "unit": "audio-s/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 utterance x 10 s (100 phonemes, 5 diffusion steps, iSTFTNet), best of 3 after 1 warm-up, "
                      "%.2f s wall" % best}


def roofline(ach, durs, avg_ms, flop, B, L_dom):
    """Dominant kernel class = the k=11, C=128, L=48001 resblock convs of the last generator stage (12 launches per
    decoder call).  `achieved` = algorithmic conv FLOPs (2*B*C_in*C_out*ks*L) / mean launch time measured with HIP
    events on the launch stream inside the timed region; `peak` = dense f16 MFMA peak / 3 products (see above), so
    `frac` is also (f16 MFMA FLOPs executed / s) / 2.5 PFLOP/s.  `traffic` = HBM bytes per launch from rocprofv3 PMC
    passes (FETCH_SIZE x2 per the gfx950 correction in MI355X_MICROARCH.md + WRITE_SIZE), read from the committed
    profiles/pmc_dominant.json; the HBM-side view (algorithmic bytes / s vs 8 TB/s) is reported beside it."""
    peak = F16_MFMA_PEAK_TFLOPS / F16S_PRODUCTS
    # every launch reads x and writes y (fp32); every second one (convs2) also reads the residual; weights are L2 resident
    alg_bytes = 2.5 * B * 128 * L_dom * 4
    traffic, note = None, "no PMC summary committed"
    if os.path.exists(PMC_FILE):
        pm = json.load(open(PMC_FILE))
        traffic, note = pm.get("hbm_bytes_per_launch"), pm.get("note")
    return {"bound": "mfma", "kernel": "conv1d_xs_kernel_o3<11,16,4,1,4> (st2_conv1d_xs: C=128, L=48001, B=32, k=11 on "
                                       "pre-activated split-f16 planes; bias/residual/statistics epilogue)",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": (ach / peak) if ach else None,
            "traffic": traffic, "traffic_note": note,
            "peak_note": "2500 TFLOP/s dense f16 MFMA / 3 products per fp32-class multiply",
            "mfma_tflops_executed": (ach * F16S_PRODUCTS) if ach else None,
            "launches_timed": len(durs), "avg_launch_ms": avg_ms, "algorithmic_flop_per_launch": flop,
            "algorithmic_bytes_per_launch": alg_bytes,
            "hbm_view": {"achieved_GBps": alg_bytes / (avg_ms * 1e-3) / 1e9 if durs else None, "peak_GBps": 8000.0}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true", help="issue every step on one stream (no front/decoder overlap)")
    ap.add_argument("--front-priority", type=int, default=-1, help="HIP stream priority of the front stream (-1 = high)")
    a = ap.parse_args()

    from _util import manifest
    from styletts2_amd import _lib, models, ops, parallel, pipeline, synth

    rank, local_rank, world = parallel.init_distributed()
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _lib.load()

    log("rank %d/%d on %s (%d host cores)" % (rank, world, torch.cuda.get_device_name(local_rank), os.cpu_count()))
    man = manifest("ljspeech")
    model = build(man)
    if rank == 0:  # seeded random weights of the reference architecture, generated once ...
        for i, k in enumerate(KEYS):
            synth.init_synthetic_(model[k], 10 + i)
    sds = {k: {n: t.clone() for n, t in model[k].state_dict().items()} for k in KEYS} if rank == 0 else None
    for k in KEYS:
        model[k].eval().to(dev)
    nbytes = parallel.broadcast_model(model, KEYS, src=0)  # ... and broadcast over RCCL/xGMI (no-op for N=1)
    sampler = models.make_sampler(model)
    log("weights ready (%d B broadcast)" % nbytes)

    B = PER_GPU_BATCH
    tokens, lengths, noise, durations = synthetic_inputs(B, 1000 + rank)
    tokens, noise = tokens.to(dev), noise.to(dev)

    # Two HIP streams: the front of step k+1 (text encoder, PL-BERT, diffusion sampler, duration / prosody predictors:
    # latency-bound small kernels) is issued on `front` and overlaps the decoder + vocoder of step k on the main
    # stream.  Every step is still one complete pass tokens -> waveform over the batch; --single-stream turns it off.
    # the front stream gets the higher priority: its small kernels then take CU slots as the decoder's workgroups
    # retire instead of queueing behind them (--front-priority 0 = equal priorities)
    front = None if a.single_stream else torch.cuda.Stream(dev, priority=a.front_priority)
    if front is not None:
        front.wait_stream(torch.cuda.current_stream(dev))

    def step():
        return pipeline.inference(model, sampler, tokens, lengths, noise, diffusion_steps=DIFFUSION_STEPS,
                                  embedding_scale=1.0, durations=durations, front_stream=front)

    for i in range(a.warmup):
        out = step()
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    # roofline leg: per-launch HIP events around the dominant kernel class (C=128, L=48001, k=11 resblock convs)
    L_dom = N_PHONEMES * FRAMES_PER_PHONEME * 2 * 60 + 1
    timer = ops.ConvTimer(ks=11, C_in=128, C_out=128, L_out=L_dom)
    ops.set_conv_timer(timer)
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    ops.set_conv_timer(None)
    dt = parallel.max_over_ranks(dt, dev)
    log("timed %d steps: %.1f ms/step" % (a.steps, dt / a.steps * 1e3))
    assert out.shape == (B, 1, int(AUDIO_S_PER_UTT * 24000)) and bool(torch.isfinite(out).all())
    ops.check_status()  # raises if a cooperative BiLSTM group timed out or a split-f16 operand left the f16 range

    if rank == 0:
        durs = timer.durations_ms()
        flop = 2.0 * B * 128 * 128 * 11 * L_dom
        avg_ms = sum(durs) / max(len(durs), 1)
        ach = flop / (avg_ms * 1e-3) / 1e12 if durs else None
        res = {
            "metric": "audio-seconds/sec (RTF^-1) end-to-end, 10 s utterance batch",
            "value": world * B * AUDIO_S_PER_UTT * a.steps / dt,
            "unit": "audio-s/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (convs/linears: f16 hi/lo split, 3 MFMA products, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "LJSpeech single-speaker, batch=32x10 s synthetic phoneme seqs per GPU, iSTFTNet, "
                                   "5 diffusion steps, 1xMI355X per rank (BASELINE.json configs[1])",
                       "global_batch": world * B, "per_gpu_batch": B, "phonemes": N_PHONEMES,
                       "audio_s_per_utt": AUDIO_S_PER_UTT, "parallelism": "utterance-sharded x%d" % world,
                       "streams": "1" if front is None else "2 (front of step k+1 overlaps decoder of step k)",
                       "weights": "seeded random init, broadcast %d B from rank 0" % nbytes},
            "roofline": roofline(ach, durs, avg_ms, flop, B, L_dom),
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(man, sds)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
