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
    import synth  # tests/synth.py: seeded synthetic weights / inputs (test + bench helper, not product code)
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
