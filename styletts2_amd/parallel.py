"""Multi-GPU execution: one process per GPU, utterances sharded statically, weights broadcast once over
RCCL/xGMI, no collectives in steady state.

The path shards by utterance: every op of the text->waveform path is per-sample (InstanceNorm, LayerNorm,
attention, the ADPM2 loop), so rank r synthesises utterances [r*n/W, (r+1)*n/W) on its own GPU and writes its own
waveforms.  The only inter-GPU traffic is the start-up broadcast of the (folded-at-use) parameters from rank 0:
~0.42 GB fp32 for the LJSpeech inference set, sent as ONE flat buffer per module so that the ring is per-link
bandwidth bound rather than latency bound (7 xGMI links x ~153 GB/s per GPU).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Static contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_module_weights(module, src=0):
    """One flat fp32 broadcast of every parameter and floating-point buffer of `module` from rank `src`.
    Returns the number of bytes sent.  No-op for world size 1."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = [p.data for p in module.parameters()] + [b for b in module.buffers() if torch.is_floating_point(b)]
    seen, uniq = set(), []
    for t in tensors:  # modules aliased under two names (diffusion.net / unet) are sent once
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            uniq.append(t)
    if not uniq:
        return 0
    flat = torch.cat([t.reshape(-1).float() for t in uniq])
    dist.broadcast(flat, src=src)
    off = 0
    for t in uniq:
        n = t.numel()
        t.copy_(flat[off:off + n].reshape(t.shape))
        off += n
    if hasattr(module, "refresh"):
        module.refresh()
    return flat.numel() * 4


def broadcast_model(model, keys, src=0):
    total = 0
    for k in keys:
        total += broadcast_module_weights(model[k], src=src)
        for m in model[k].modules():
            if hasattr(m, "refresh"):
                m.refresh()
    return total


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device):
    """[value of rank 0, ..., value of rank W-1] on every rank (a list of one for world size 1)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def broadcast_calibration(model, device, src=0):
    """Rank `src` calibrated the split-f16 operand scales (pipeline.calibrate); every rank installs the same table so that all
    shards compute bit-identical functions of their inputs.  One small broadcast at start-up (a few hundred floats), none in
    steady state.  Returns the number of scales sent (0 for world size 1)."""
    from . import pipeline
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    engs = pipeline.model_engines(model, device)
    keys = sorted(engs)
    sizes = [len(engs[k].calibration()) for k in keys]
    flat = torch.zeros((sum(sizes),), dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        flat = torch.tensor([s for k in keys for s in engs[k].calibration_scales()], dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    host = flat.cpu().tolist()
    off = 0
    for k, n in zip(keys, sizes):
        engs[k].set_calibration(host[off:off + n])
        off += n
    return len(host)
