"""N>1 path on CPU: world_size-2 gloo processes exercise the start-up weight broadcast and the static utterance
sharding (the data path itself has no collective)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from styletts2_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = parallel.init_distributed(backend="gloo")
    from styletts2_amd.layers import AdaINResBlock1Params
    torch.manual_seed(100 + rank)  # ranks start from different weights
    blk = AdaINResBlock1Params(8, 3, (1, 3), 16)
    before = torch.cat([p.detach().reshape(-1) for p in blk.parameters()]).clone()
    nbytes = parallel.broadcast_module_weights(blk, src=0)
    after = torch.cat([p.detach().reshape(-1) for p in blk.parameters()])
    gathered = [torch.zeros_like(after) for _ in range(w)]
    dist.all_gather(gathered, after)
    lo, hi = parallel.shard_range(7, r, w)
    t = parallel.max_over_ranks(float(rank + 1), "cpu")
    parallel.barrier()
    q.put((rank, nbytes, bool(torch.equal(gathered[0], gathered[1])), bool(torch.equal(before, after)), (lo, hi), t))
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, same0, unchanged0, sh0, t0), (r1, n1, same1, unchanged1, sh1, t1) = res
    assert n0 == n1 > 0 and same0 and same1
    assert unchanged0 and not unchanged1  # rank 0 is the source; rank 1 was overwritten
    assert sh0 == (0, 4) and sh1 == (4, 7)
    assert t0 == t1 == 2.0


def _calibration_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    from _cpu_backend import cpu_backend
    from _util import manifest
    from benchdata import synth
    from styletts2_amd import models, pipeline
    man = manifest("ljspeech")
    model = models.build_model(models.recursive_munch(man["config"]), None, None, models.load_plbert(man["plbert"]))
    for i, k in enumerate(["decoder", "diffusion", "predictor", "text_encoder", "bert_encoder", "bert"]):
        synth.init_synthetic_(model[k], 10 + i)
        model[k].eval()
    dev = torch.device("cpu")
    with cpu_backend():
        engs = pipeline.model_engines(model, dev)
        if rank == 0:  # rank 0 "calibrated": a table of distinct powers of two per engine; rank 1 still runs by rule
            for j, k in enumerate(sorted(engs)):
                engs[k].set_calibration([2.0 ** (1 + (i + j) % 9) for i in range(len(engs[k].calibration()))])
        before = pipeline.calibration_state(model, dev)
        n_sent = parallel.broadcast_calibration(model, dev)
        after = pipeline.calibration_state(model, dev)
        gens = {k: e.calib_gen for k, e in engs.items()}
    parallel.barrier()
    q.put((rank, n_sent, before, after, gens))
    dist.destroy_process_group()


def test_calibration_table_broadcast_world2():
    """parallel.broadcast_calibration: the operand-scale table rank 0 measured (pipeline.calibrate) is installed on every rank --
    one small broadcast at start-up, so that all shards compute bit-identical functions of their inputs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_calibration_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, n0, before0, after0, _), (_, n1, before1, after1, gens1) = res
    assert n0 == n1 == sum(len(v) for v in after0.values()) > 100
    assert {"front", "decoder"} <= set(after0) and after0 == before0   # the source keeps its table
    assert all(x == 0.0 for v in before1.values() for x in v)          # rank 1 ran by rule ...
    assert after1 == after0                                            # ... and now holds rank 0's table, value for value
    assert all(g >= 1 for g in gens1.values())                         # recorded graphs of rank 1 are stale (calib_gen moved)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 256):
        for w in (1, 2, 4, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _bench_dry_run(cmd):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line, got %r" % out.stdout
    return json.loads(lines[0])


def test_bench_self_spawns_n_ranks():
    """`python bench.py --gpus N` (how the driver may call it) must run N ranks, not one: with no torch.distributed.run
    environment the script spawns one process per GPU itself; --dry-run exercises launch, rendezvous (gloo here), barrier
    and the max-over-ranks reduction without compute."""
    import sys
    res = _bench_dry_run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"])
    assert res["n_gpus"] == 2 and abs(res["max_over_ranks"] - 0.002) < 1e-9  # rank 1's value won the MAX reduction
    # the fields that make a SCALE record self-explanatory: every rank's own time, the start-up broadcast's cost
    assert res["per_rank_ms_per_step"] == [1.0, 2.0]
    assert res["broadcast_bytes"] == (8 * 8 + 8) * 4 and res["broadcast_s"] >= 0.0


def test_bench_under_torch_distributed_run():
    import sys
    res = _bench_dry_run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "2",
                          "--dry-run"])
    assert res["n_gpus"] == 2 and res["per_rank_ms_per_step"] == [1.0, 2.0] and res["broadcast_bytes"] == 288
