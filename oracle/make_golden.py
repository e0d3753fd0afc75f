"""Generates the committed fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference, via oracle/ref_harness.py) in this container.  Not runnable on the GPU box.

    python -m oracle.make_golden manifests      # state_dict key/shape manifests
    python -m oracle.make_golden vectors        # golden tap-point vectors

Outputs are small (manifests: JSON; vectors: .npz of fp32 arrays at reduced sequence lengths).
"""
import json
import os
import sys

import numpy as np
import torch

from oracle import ref_harness as RH

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
MANIFESTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchdata", "manifests")
HOT_MODULES = ["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert", "style_encoder",
               "predictor_encoder"]
HIFIGAN_OVERRIDE = {"multispeaker": True,
                    "decoder": {"type": "hifigan", "upsample_rates": [10, 5, 3, 2],
                                "upsample_kernel_sizes": [20, 10, 6, 4]}}


def libritts_overrides():
    cfg = RH.load_config("config_libritts.yml")["model_params"]
    return {"multispeaker": cfg["multispeaker"], "decoder": cfg["decoder"]}


def istftnet_decoder_override():
    """BASELINE.json configs[3]: LibriTTS (multispeaker) with the iSTFTNet decoder.  The reference ships LibriTTS only
    with `decoder.type: hifigan` (Configs/config_libritts.yml:49-55) but the two switches are independent in
    build_model (models.py:617-633 vs :643-651) and the LibriTTS notebook itself tests `decoder.type == "hifigan"`
    before its one-frame shift: the combination is built by replacing the decoder block of config_libritts.yml with
    that of Configs/config.yml:49-57."""
    return {"decoder": RH.load_config("config.yml")["model_params"]["decoder"]}


CONFIGS = (("ljspeech", "config.yml", None), ("libritts", "config_libritts.yml", None),
           ("libritts_istftnet", "config_libritts.yml", "istftnet"))


def manifests():
    os.makedirs(MANIFESTS, exist_ok=True)
    for tag, cfgname, ov in CONFIGS:
        model, args, cfg = RH.build_reference_model(cfgname, seed=0,
                                                    overrides=istftnet_decoder_override() if ov else None,
                                                    replace_keys=("decoder",) if ov else ())
        man = {"config": cfg["model_params"], "plbert": RH.plbert_config(), "modules": {}}
        for key in HOT_MODULES:
            sd = model[key].state_dict()
            man["modules"][key] = {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                                       "std": float(v.float().std()) if v.numel() > 1 else 0.0,
                                       "mean": float(v.float().mean())}
                                   for k, v in sd.items()}
        path = os.path.join(MANIFESTS, "manifest_%s.json" % tag)
        with open(path, "w") as f:
            json.dump(man, f, indent=0, sort_keys=True)
        print(path, {k: len(v) for k, v in man["modules"].items()})


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "manifests"
    if what == "manifests":
        manifests()
    elif what == "vectors":
        from oracle import golden_vectors
        golden_vectors.main()
