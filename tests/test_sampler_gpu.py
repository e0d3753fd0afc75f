"""Style-diffusion sampler: HIP engine vs the CPU oracle with replayed noise."""
import pytest
import torch

from _util import manifest
from oracle import st2_oracle as O
from styletts2_amd import models
from benchdata import synth  # seeded synthetic weights / inputs (test + bench helper, not product code)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _diffusion(tag, seed=2):
    man = manifest(tag)
    args = models.recursive_munch(man["config"])

    class _B:  # only bert.config.{hidden_size,max_position_embeddings} are read by the builder
        config = type("C", (), dict(hidden_size=768, max_position_embeddings=512))()
    tcls = models.StyleTransformer1d if args.multispeaker else models.Transformer1d
    tr = tcls(channels=args.style_dim * 2, context_embedding_features=768, context_features=args.style_dim * 2,
              embedding_max_length=512, **args.diffusion.transformer)
    diff = models.AudioDiffusionConditional(tr, sigma_data=args.diffusion.dist.sigma_data).eval()
    synth.init_synthetic_(diff, seed)
    return man, diff


@pytest.mark.parametrize("tag,B,N,steps,scale", [("ljspeech", 2, 37, 5, 1.0), ("ljspeech", 3, 100, 5, 1.5),
                                                 ("libritts", 2, 64, 10, 1.0), ("libritts", 1, 130, 5, 2.0),
                                                 ("ljspeech", 1, 512, 3, 1.0),
                                                 ("libritts", 4, 80, 4, 1.5)])  # B * N >= 256: merged q / kv GEMMs
def test_sampler_matches_oracle_per_step(tag, B, N, steps, scale):
    man, diff = _diffusion(tag)
    sd = O.sub(diff.state_dict(), "unet")
    g = torch.Generator().manual_seed(N)
    noise = torch.randn(B, 1, 256, generator=g)
    emb = torch.randn(B, N, 768, generator=g)
    feats = torch.randn(B, 256, generator=g) if man["config"]["multispeaker"] else None
    step_noise = torch.randn(steps - 1, B, 1, 256, generator=g)
    to, te = {}, {}
    ref = O.sample_style(sd, noise, emb, steps, step_noise, sigma_data=0.2, features=feats, embedding_scale=scale,
                         taps=to)
    diff = diff.to(DEV)
    sampler = models.DiffusionSampler(diff.diffusion, sampler=models.ADPM2Sampler(),
                                      sigma_schedule=models.KarrasSchedule(sigma_min=0.0001, sigma_max=3.0, rho=9.0),
                                      clamp=False)
    kw = dict(embedding=emb.to(DEV), embedding_scale=scale, num_steps=steps, step_noise=step_noise.to(DEV), taps=te)
    if feats is not None:
        kw["features"] = feats.to(DEV)
    out = sampler(noise.to(DEV), **kw)
    torch.cuda.synchronize()
    assert out.shape == ref.shape == (B, 1, 256)
    for k in to:  # s_pred after every ADPM2 step (tap-point protocol i)
        assert (te[k].cpu() - to[k]).abs().max().item() < 2e-5, k
    assert (out.cpu() - ref).abs().max().item() < 2e-5


def test_denoiser_forward_signature_and_schedule():
    man, diff = _diffusion("ljspeech")
    sd = O.sub(diff.state_dict(), "unet")
    g = torch.Generator().manual_seed(1)
    x, emb = torch.randn(2, 1, 256, generator=g), torch.randn(2, 20, 768, generator=g)
    t = torch.full((2,), -0.3)
    ref = O.denoiser_net(sd, x, t, emb)
    diff = diff.to(DEV)
    out = diff.unet(x.to(DEV), t.to(DEV), embedding=emb.to(DEV))
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    # Karras schedule and ADPM2 sigmas are host values in the reference's arithmetic (App. A.4)
    s5 = models.KarrasSchedule(1e-4, 3.0, 9.0)(5)
    assert torch.equal(s5, O.karras_schedule(5))
    assert abs(float(s5[1]) - 0.557915) < 1e-5 and float(s5[-1]) == 0.0
    up, down, mid = models.ADPM2Sampler().get_sigmas(s5[0], s5[1])
    assert abs(up - 0.548182) < 1e-5 and abs(down - 0.103756) < 1e-5 and abs(mid - 1.55188) < 1e-4
    with pytest.raises(AssertionError):
        diff.unet(x.to(DEV), t.to(DEV))  # embedding is mandatory


@pytest.mark.parametrize("tag,B,N,steps,scale", [("ljspeech", 2, 41, 5, 1.0), ("libritts", 3, 100, 4, 1.5)])
def test_graphed_sampler_replays_match_eager(tag, B, N, steps, scale):
    """GraphedSampler (hipGraph capture of a whole sampler run, BASELINE.json configs[4]): replays with NEW inputs of a
    captured signature are bitwise the eager results; a second signature gets its own graph; without explicit step
    noise two replays draw different noise."""
    man, diff = _diffusion(tag)
    diff = diff.to(DEV)
    eager = models.DiffusionSampler(diff.diffusion, sampler=models.ADPM2Sampler(),
                                    sigma_schedule=models.KarrasSchedule(1e-4, 3.0, 9.0), clamp=False)
    graphed = models.GraphedSampler(eager)
    multi = man["config"]["multispeaker"]
    g = torch.Generator().manual_seed(7)

    def case(n):
        kw = dict(embedding=torch.randn(B, n, 768, generator=g).to(DEV), embedding_scale=scale, num_steps=steps,
                  step_noise=torch.randn(steps - 1, B, 1, 256, generator=g).to(DEV))
        if multi:
            kw["features"] = torch.randn(B, 256, generator=g).to(DEV)
        return torch.randn(B, 1, 256, generator=g).to(DEV), kw

    for n in (N, N, N + 3, N):  # capture, replay, second signature, replay of the first again
        noise, kw = case(n)
        ref = eager(noise, **kw)
        out = graphed(noise, **kw)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    assert len(graphed._graphs) == 2
    noise, kw = case(N)
    kw.pop("step_noise")
    a, b = graphed(noise, **kw), graphed(noise, **kw)
    assert bool(torch.isfinite(a).all()) and not torch.equal(a, b)
