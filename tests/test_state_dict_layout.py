"""Drop-in boundary: every hot-path module built by `build_model` has the reference's state_dict layout, key for
key and shape for shape (manifests generated from the reference by oracle/make_golden.py)."""
import pytest

from _util import manifest
from styletts2_amd import models

HOT = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert", "style_encoder", "predictor_encoder"]


@pytest.mark.parametrize("tag", ["ljspeech", "libritts"])
def test_state_dict_layout_matches_reference(tag):
    man = manifest(tag)
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    assert set(model.keys()) == {"bert", "bert_encoder", "predictor", "decoder", "text_encoder", "predictor_encoder",
                                 "style_encoder", "diffusion", "text_aligner", "pitch_extractor", "mpd", "msd", "wd"}
    for key in HOT:
        mine = {k: list(v.shape) for k, v in model[key].state_dict().items()}
        ref = {k: v["shape"] for k, v in man["modules"][key].items()}
        assert mine == ref, (key, set(mine) ^ set(ref))
    for key in model:  # every entry quacks like nn.Module (notebooks call .eval()/.to() on all of them)
        model[key].eval()
    assert model.diffusion.diffusion.alias == "k" and model.diffusion.diffusion.net is model.diffusion.unet
    model.diffusion.diffusion.sigma_data = 0.19  # mutable float attribute, not a parameter
    with pytest.raises(NotImplementedError):
        model.mpd(None)


def test_checkpoint_roundtrip_with_module_prefix(tmp_path):
    import torch
    from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)
    man = manifest("ljspeech")
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    synth.init_synthetic_(model.text_encoder, 5)
    synth.init_synthetic_(model.bert_encoder, 6)
    ckpt = {"net": {"text_encoder": {"module." + k: v for k, v in model.text_encoder.state_dict().items()},
                    "bert_encoder": model.bert_encoder.state_dict()}, "epoch": 3, "iters": 7}
    path = str(tmp_path / "epoch_2nd_00003.pth")
    torch.save(ckpt, path)
    fresh = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    fresh, _, epoch, iters = models.load_checkpoint(fresh, None, path, ignore_modules=["bert_encoder"])
    assert (epoch, iters) == (0, 0)
    for k, v in model.text_encoder.state_dict().items():
        assert torch.equal(v, fresh.text_encoder.state_dict()[k])
    assert not torch.equal(model.bert_encoder.weight, fresh.bert_encoder.weight)  # ignored module untouched


def test_modules_pickle_and_deepcopy_without_their_process_local_caches():
    """Packed weights, st2_engine handles (ctypes pointers) and the owner weakref of `DurationEncoder` are process-local
    caches (styletts2_amd/layers.py `transient_state`): `torch.save(module)`, `pickle` and `copy.deepcopy` drop them, a copy
    of the predictor owns its own text_encoder, and the copy's parameters equal the original's (advisor, round 3)."""
    import copy
    import io

    import torch

    from benchdata import manifest, synth
    from styletts2_amd import models
    man = manifest("ljspeech")
    args = models.recursive_munch(man["config"])
    model = models.build_model(args, None, None, models.load_plbert(man["plbert"]))
    for i, k in enumerate(["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]):
        m = model[k]
        synth.init_synthetic_(m, 10 + i)
        # what a GPU run would have cached, where the product caches it: must not travel
        holder = m.diffusion.net if k == "diffusion" else m
        holder.__dict__["_engine"] = ("stamp", object())
        c = copy.deepcopy(m)
        others = [c]
        if k != "bert":  # PL-BERT's class is created inside build_plbert (lazy `transformers` import): it travels as a
            buf = io.BytesIO()  # state_dict, like the reference's own Utils/PLBERT/util.py:load_plbert does
            torch.save(m, buf)
            buf.seek(0)
            others.append(torch.load(buf, weights_only=False))
        for other in others:
            oh = other.diffusion.net if k == "diffusion" else other
            assert oh.__dict__.get("_engine") is None and oh.__dict__.get("_pk") is None
            sa, sb = m.state_dict(), other.state_dict()
            assert sa.keys() == sb.keys() and all(torch.equal(sa[n], sb[n]) for n in sa)
        del holder.__dict__["_engine"]
    p2 = copy.deepcopy(model.predictor)
    assert p2.text_encoder.__dict__["_owner"]() is p2
    assert model.predictor.text_encoder.__dict__["_owner"]() is model.predictor
