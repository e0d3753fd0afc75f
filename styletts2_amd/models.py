"""`build_model` / `load_checkpoint`: the reference's model-builder API (models.py:614-713) over the MI355X engine.

    config = yaml.safe_load(open("Configs/config.yml"))
    model = build_model(recursive_munch(config["model_params"]), text_aligner, pitch_extractor, plbert)

returns a Munch with the reference's keys.  Hot-path entries (decoder, diffusion, predictor, text_encoder, bert,
bert_encoder, and the reference-audio style encoders style_encoder / predictor_encoder) are engine-backed nn.Modules
with the reference's call signatures and state_dict layouts; the training-only entries (discriminators, aligner, pitch
extractor) are explicit placeholders that raise if called ("Scope").
"""
import torch
import torch.nn as nn

from .decoder import Decoder
from .diffusion import (GraphedSampler, ADPM2Sampler, AudioDiffusionConditional, DiffusionSampler, KarrasSchedule,  # noqa: F401
                        StyleTransformer1d, Transformer1d)
from .style import StyleEncoder
from .text import EngineLinear, ProsodyPredictor, TextEncoder, build_plbert
from .utils import Munch, recursive_munch  # noqa: F401
from .weights import strip_module_prefix

PLBERT_DEFAULTS = dict(vocab_size=178, hidden_size=768, num_attention_heads=12, intermediate_size=2048,
                       max_position_embeddings=512, num_hidden_layers=12, dropout=0.1)  # Utils/PLBERT/config.yml:23-30


class OutOfScope(nn.Module):
    """Placeholder for a reference component outside the inference hot path."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def forward(self, *a, **k):
        raise NotImplementedError("%s is not part of the MI355X inference engine (see DESIGN.md, scope)" % self.what)


def load_plbert(plbert_params=None):
    """Utils/PLBERT/util.py:15-42 builds ALBERT from a config.yml and loads the newest step_*.t7; offline there is
    no checkpoint, so this returns the randomly initialised architecture (weights via load_state_dict)."""
    return build_plbert(dict(PLBERT_DEFAULTS, **(plbert_params or {})))


def build_model(args, text_aligner=None, pitch_extractor=None, bert=None):
    """models.py:614-694."""
    assert args.decoder.type in ["istftnet", "hifigan"], "Decoder type unknown"
    dc = args.decoder
    kw = dict(dim_in=args.hidden_dim, style_dim=args.style_dim, dim_out=args.n_mels,
              resblock_kernel_sizes=dc.resblock_kernel_sizes, upsample_rates=dc.upsample_rates,
              upsample_initial_channel=dc.upsample_initial_channel,
              resblock_dilation_sizes=dc.resblock_dilation_sizes, upsample_kernel_sizes=dc.upsample_kernel_sizes,
              kind=dc.type)
    if dc.type == "istftnet":
        kw.update(gen_istft_n_fft=dc.gen_istft_n_fft, gen_istft_hop_size=dc.gen_istft_hop_size)
    decoder = Decoder(**kw)
    text_encoder = TextEncoder(channels=args.hidden_dim, kernel_size=5, depth=args.n_layer, n_symbols=args.n_token)
    predictor = ProsodyPredictor(style_dim=args.style_dim, d_hid=args.hidden_dim, nlayers=args.n_layer,
                                 max_dur=args.max_dur, dropout=args.dropout)
    if bert is None:
        bert = load_plbert()
    tcls = StyleTransformer1d if args.multispeaker else Transformer1d
    transformer = tcls(channels=args.style_dim * 2, context_embedding_features=bert.config.hidden_size,
                       context_features=args.style_dim * 2,
                       embedding_max_length=bert.config.max_position_embeddings, **args.diffusion.transformer)
    diffusion = AudioDiffusionConditional(transformer, sigma_data=args.diffusion.dist.sigma_data,
                                          embedding_mask_proba=args.diffusion.embedding_mask_proba)
    return Munch(
        bert=bert,
        bert_encoder=EngineLinear(bert.config.hidden_size, args.hidden_dim),
        predictor=predictor,
        decoder=decoder,
        text_encoder=text_encoder,
        # reference-audio style encoders (models.py:639-640): acoustic / prosodic halves of ref_s, see style.py
        predictor_encoder=StyleEncoder(dim_in=args.dim_in, style_dim=args.style_dim, max_conv_dim=args.hidden_dim),
        style_encoder=StyleEncoder(dim_in=args.dim_in, style_dim=args.style_dim, max_conv_dim=args.hidden_dim),
        diffusion=diffusion,
        text_aligner=text_aligner if text_aligner is not None else OutOfScope("text_aligner (training only)"),
        pitch_extractor=pitch_extractor if pitch_extractor is not None else OutOfScope("pitch_extractor (training only)"),
        mpd=OutOfScope("mpd (training only)"),
        msd=OutOfScope("msd (training only)"),
        wd=OutOfScope("wd (training only)"),
    )


def load_checkpoint(model, optimizer, path, load_only_params=True, ignore_modules=()):
    """models.py:696-713: `torch.load(path)['net'][key]` state_dicts, `module.` prefixes tolerated, strict=False."""
    state = torch.load(path, map_location="cpu")
    params = state["net"]
    for key in model:
        if key in params and key not in ignore_modules and not isinstance(model[key], OutOfScope):
            model[key].load_state_dict(strip_module_prefix(params[key]), strict=False)
    for key in model:
        model[key].eval()
    if not load_only_params and optimizer is not None:
        optimizer.load_state_dict(state["optimizer"])
        return model, optimizer, state["epoch"], state["iters"]
    return model, optimizer, 0, 0


def make_sampler(model, clamp=False, graph=False):
    """The sampler every notebook builds (Demo/Inference_LJSpeech.ipynb:234-239); graph=True wraps it in
    `GraphedSampler` (one hipGraph replay per run instead of ~300 kernel launches)."""
    sampler = DiffusionSampler(model.diffusion.diffusion, sampler=ADPM2Sampler(),
                               sigma_schedule=KarrasSchedule(sigma_min=0.0001, sigma_max=3.0, rho=9.0), clamp=clamp)
    return GraphedSampler(sampler) if graph else sampler
