"""a checkpoint in the reference's FULL layout through `models.load_checkpoint` / the notebooks'
loading loop, and one inference from it.

What `train_second.py:774-789` writes: {'net': {key: model[key].state_dict() for all 13 keys of build_model}, 'optimizer',
'iters', 'val_loss', 'epoch'} -- with `module.` prefixes on the modules that were wrapped in nn.DataParallel -- and, next
to it, the training config with `model_params.diffusion.dist.sigma_data` overwritten by the estimated value.  What
`models.py:696-713` / Demo/Inference_LibriTTS.ipynb do with it: load every key the model has, tolerate the prefixes,
`strict=False`, eval().  The training-only entries (text_aligner, pitch_extractor, mpd, msd, wd) are in the file and must
be skipped without loading anything; `sigma_data` must come back from the saved config and reach the sampler's
pre-conditioning.  No real checkpoint exists offline: the file is synthesised in the reference's layout (the hot-path
state_dict layouts themselves are pinned key for key against the reference in test_state_dict_layout.py)."""
import os

import pytest
import torch
import yaml

from _util import manifest
from styletts2_amd import models
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)

HOT = ["bert", "bert_encoder", "predictor", "decoder", "text_encoder", "diffusion"]
STYLE = ["predictor_encoder", "style_encoder"]
TRAIN_ONLY = ["text_aligner", "pitch_extractor", "mpd", "msd", "wd"]
SIGMA_DATA = 0.1734  # an "estimated" value, different from the config default 0.2


def _write_checkpoint(tmp_path, tag="libritts"):
    """(model the file was written from, checkpoint path, saved-config path)."""
    man = manifest(tag)
    args = models.recursive_munch(man["config"])
    src = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    for i, k in enumerate(HOT):
        synth.init_synthetic_(src[k], 50 + i)
    for i, k in enumerate(STYLE):
        synth.init_spectral_norm_(src[k], 70 + i)
    net = {}
    for k in HOT + STYLE:
        sd = src[k].state_dict()
        # train_second.py wraps every module in DataParallel (MyDataParallel): the published checkpoints carry `module.`
        net[k] = {("module." + n): v.clone() for n, v in sd.items()} if k != "bert_encoder" else dict(sd)
    g = torch.Generator().manual_seed(1)
    for k in TRAIN_ONLY:  # present in the file with their own (here: made-up) tensors; never loaded by the engine
        net[k] = {"module.some.weight": torch.randn(4, 4, generator=g), "module.some.bias": torch.randn(4, generator=g)}
    assert len(net) == 13
    ckpt = {"net": net, "optimizer": {"state": {}, "param_groups": []}, "iters": 4242, "val_loss": 0.5, "epoch": 17}
    path = os.path.join(str(tmp_path), "epoch_2nd_00017.pth")
    torch.save(ckpt, path)
    cfg = {"model_params": yaml.safe_load(yaml.safe_dump(man["config"])), "log_dir": "Models/LibriTTS"}
    cfg["model_params"]["diffusion"]["dist"]["sigma_data"] = SIGMA_DATA  # train_second.py:786-789
    cfg_path = os.path.join(str(tmp_path), "config_libritts.yml")
    with open(cfg_path, "w") as f:
        yaml.dump(cfg, f, default_flow_style=True)
    return src, man, path, cfg_path


def _load(man, path, cfg_path):
    config = yaml.safe_load(open(cfg_path))  # the notebooks' own first step
    args = models.recursive_munch(config["model_params"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    model, _, epoch, iters = models.load_checkpoint(model, None, path, load_only_params=True)
    return model, epoch, iters


def test_full_layout_checkpoint_loads(tmp_path):
    src, man, path, cfg_path = _write_checkpoint(tmp_path)
    model, epoch, iters = _load(man, path, cfg_path)
    assert (epoch, iters) == (0, 0)  # load_only_params=True, models.py:707-711
    for k in HOT + STYLE:
        want, got = src[k].state_dict(), model[k].state_dict()
        assert list(want) == list(got), k
        for n in want:
            assert torch.equal(want[n], got[n]), (k, n)
        assert not model[k].training
    for k in TRAIN_ONLY:  # still the explicit placeholders, nothing was loaded into them
        assert isinstance(model[k], models.OutOfScope) and len(model[k].state_dict()) == 0
    # sigma_data: read back from the SAVED config, into the EDM pre-conditioning the sampler uses
    assert model.diffusion.diffusion.sigma_data == SIGMA_DATA
    assert models.make_sampler(model).diffusion.sigma_data == SIGMA_DATA
    # the notebooks' loading loop (Demo/Inference_LibriTTS.ipynb: load_state_dict, on failure strip `module.` and retry
    # with strict=False) works on the same file
    fresh = models.build_model(models.recursive_munch(yaml.safe_load(open(cfg_path))["model_params"]), None, None,
                               models.load_plbert(man["plbert"]))
    params = torch.load(path, map_location="cpu")["net"]
    for key in fresh:
        if key in params and not isinstance(fresh[key], models.OutOfScope):
            try:
                fresh[key].load_state_dict(params[key])
            except Exception:
                fresh[key].load_state_dict({k[7:]: v for k, v in params[key].items()}, strict=False)
    for k in HOT + STYLE:
        for n, v in src[k].state_dict().items():
            assert torch.equal(v, fresh[k].state_dict()[n]), (k, n)
    # optimizer / epoch / iters restore path (load_only_params=False), models.py:703-706
    class _Opt:
        def load_state_dict(self, sd):
            self.sd = sd
    opt = _Opt()
    _, opt2, epoch, iters = models.load_checkpoint(fresh, opt, path, load_only_params=False)
    assert (epoch, iters) == (17, 4242) and opt2.sd == {"state": {}, "param_groups": []}


@pytest.mark.gpu
def test_inference_from_full_layout_checkpoint_matches_oracle(tmp_path):
    """Text -> waveform on the GPU from the loaded file, against the oracle evaluated on the file's own tensors (prefixes
    stripped) and the saved config's sigma_data: style vector per the ADPM2 sampler, F0, and the waveform with the
    oracle's harmonic features injected (tap-point protocol)."""
    from oracle import st2_oracle as O
    from styletts2_amd import pipeline
    from _util import rms
    src, man, path, cfg_path = _write_checkpoint(tmp_path)
    model, _, _ = _load(man, path, cfg_path)
    cfg = yaml.safe_load(open(cfg_path))["model_params"]
    raw = torch.load(path, map_location="cpu")["net"]
    sds = {k: {(n[7:] if n.startswith("module.") else n): v for n, v in raw[k].items()} for k in HOT}
    g = torch.Generator().manual_seed(0)
    B, N, steps = 2, 9, 4
    tokens = torch.randint(1, 178, (B, N), generator=g)
    tokens[:, 0] = 0
    lengths = torch.LongTensor([N] * B)
    noise = torch.randn(B, 1, 256, generator=g)
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    ref_s = torch.randn(B, 256, generator=g)
    dur = torch.full((B, N), 2, dtype=torch.long)
    sine_noise = torch.randn(B, 600 * 2 * N, 9, generator=g)
    to = {}
    ref = O.inference(sds, cfg, man["plbert"], tokens, lengths, noise, step_noise, sine_noise, diffusion_steps=steps,
                      ref_s=ref_s, durations=dur, taps=to)
    # the sampler result depends on sigma_data: the config default would give another style vector
    tother = {}
    O.front(sds, dict(cfg, diffusion=dict(cfg["diffusion"], dist=dict(cfg["diffusion"]["dist"], sigma_data=0.2))),
            man["plbert"], tokens, lengths, noise, step_noise, diffusion_steps=steps, ref_s=ref_s, durations=dur,
            taps=tother)
    dev = "cuda"
    for k in HOT:
        model[k].to(dev)
    sampler = models.make_sampler(model)
    te = {}
    out = pipeline.inference(model, sampler, tokens.to(dev), lengths, noise.to(dev), diffusion_steps=steps,
                             ref_s=ref_s.to(dev), durations=dur, step_noise=step_noise.to(dev),
                             sine_noise=sine_noise.to(dev), taps=te)
    assert out.shape == ref.shape and bool(torch.isfinite(out).all())
    e_s = (te["s_pred"].cpu() - to["s_pred"]).abs().max().item()
    assert e_s < 1e-4, e_s
    assert (to["s_pred"] - tother["s_pred"]).abs().max().item() > 1e-3  # sigma_data did matter
    assert (te["F0"].cpu() - to["F0"]).abs().max().item() < 1e-4 * to["F0"].abs().max().item()
    wave = model.decoder(to["asr"].to(dev), to["F0"].to(dev), to["N"].to(dev), to["s_mixed"][:, :128].contiguous().to(dev),
                         noise=sine_noise.to(dev), har=to["har"].to(dev))
    assert rms(wave.cpu() - ref) < 1e-4
    # and the product path proper (C++ plans, no taps) agrees with the tap run on everything deterministic
    out2 = pipeline.inference(model, sampler, tokens.to(dev), lengths, noise.to(dev), diffusion_steps=steps,
                              ref_s=ref_s.to(dev), durations=dur, step_noise=step_noise.to(dev),
                              sine_noise=sine_noise.to(dev))
    assert out2.shape == out.shape and bool(torch.isfinite(out2).all())
