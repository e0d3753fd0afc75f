"""The phase schedule of the warp-specialised fused conv (styletts2_amd/csrc/st2_conv1d_f16s_ws.h), replayed on the CPU.

The kernel's two roles meet at one workgroup barrier per phase and share two LDS chunk buffers and two parameter-table slots;
which chunk may be staged in which phase is a few lines of integer arithmetic (`consumed_at`, `wait_for_buffer`, the load
cursor, the `b & 1` table slot).  This model runs exactly that arithmetic for one workgroup -- producers and consumers as
coroutines that yield at every barrier -- and checks what the GPU parity tests can only show indirectly: both roles execute the
same number of barriers (no hang), no buffer is overwritten before its chunk was consumed or read before it was staged, and
the table slot a chunk is activated with holds its batch item and is not rewritten in the same phase."""
import itertools

import pytest


def producers(tiles_b, nchunk, lds, log, depth=2):
    """tiles_b[i] = batch item of the workgroup's i-th tile.  Mirrors the producer branch of conv1d_f16s_ws_kernel; `depth` = the
    kernel's 2 (LDS chunk buffers), other values only to show that the checks bite."""
    ntile = len(tiles_b)
    total = ntile * nchunk
    nphase = ntile * (nchunk + 1)
    cursor = [0, 0]  # load cursor (tile, chunk): parks on the last chunk of the range
    sets = [None, None]  # what each register set holds: (global chunk index or None when parked past the end, batch item)
    tp = {"b": None}
    tab_b = [None]
    loaded = [0]

    def load_set(s):
        t, c = cursor
        g = t * nchunk + c
        sets[s] = (g if loaded[0] == g else None, tiles_b[t])  # a parked cursor re-reads the last chunk: never activated
        loaded[0] = max(loaded[0], g + 1)
        tp["b"] = tiles_b[t]
        if c + 1 < nchunk:
            cursor[1] = c + 1
        elif t + 1 < ntile:
            cursor[0], cursor[1] = t + 1, 0

    def activate_set(s, buf, phase):
        g, b = sets[s]
        assert g is not None, "a register set that holds no fresh chunk is being staged"
        slot = lds["table"][b & 1]
        assert slot["b"] == b, "chunk %d (batch item %d) activated with the table of batch item %s" % (g, b, slot["b"])
        assert slot["written_in_phase"] != phase, "table slot rewritten in the phase that uses it"
        assert lds["buf"][buf]["state"] == "free", "chunk %d staged into buffer %d which still holds chunk %s" % (
            g, buf, lds["buf"][buf]["chunk"])
        lds["buf"][buf] = {"state": "full", "chunk": g, "phase": phase}
        log.append(("stage", g, phase))

    def table_update(phase):
        if tp["b"] != tab_b[0]:
            slot = tp["b"] & 1
            assert lds["table_in_use"].get(phase) != slot, "table slot %d rewritten while a chunk is activated with it" % slot
            lds["table"][slot] = {"b": tp["b"], "written_in_phase": phase}
            tab_b[0] = tp["b"]

    def produce(par, phase):
        load_set(par ^ 1)
        lds["table_in_use"][phase] = sets[par][1] & 1
        activate_set(par, par, phase)
        table_update(phase)

    def consumed_at(ph):
        return ph // (nchunk + 1) * nchunk + ph % (nchunk + 1)

    load_set(0)
    table_update(-2)
    yield "barrier"
    produce(0, -1)
    yield "barrier"
    ph = 0
    p = 1
    while p < total:
        for par in (1, 0):
            if p >= total:
                break
            assert p & 1 == par, "register-set parity must follow the chunk index"
            while p - consumed_at(ph) >= depth:
                yield "barrier"
                ph += 1
            produce(par, ph)
            yield "barrier"
            ph += 1
            p += 1
    while ph < nphase:
        yield "barrier"
        ph += 1


def consumers(tiles_b, nchunk, lds, log):
    ntile = len(tiles_b)
    yield "barrier"
    yield "barrier"
    g = 0
    phase = 0
    for t in range(ntile):
        for c in range(nchunk):
            buf = lds["buf"][g & 1]
            assert buf["state"] == "full" and buf["chunk"] == g, "chunk %d read from a buffer holding %s (%s)" % (
                g, buf["chunk"], buf["state"])
            assert buf["phase"] < phase, "chunk %d read in the phase it is staged in" % g
            lds["reading"] = g & 1
            log.append(("consume", g, phase))
            yield "barrier"
            lds["buf"][g & 1] = {"state": "free", "chunk": None, "phase": None}
            lds["reading"] = None
            g += 1
            phase += 1
        log.append(("epilogue", t, phase))
        yield "barrier"
        phase += 1


def run_workgroup(tiles_b, nchunk):
    lds = {"buf": [{"state": "free", "chunk": None, "phase": None} for _ in range(2)],
           "table": [{"b": None, "written_in_phase": None} for _ in range(2)], "table_in_use": {}, "reading": None}
    log = []
    pr, co = producers(tiles_b, nchunk, lds, log), consumers(tiles_b, nchunk, lds, log)
    barriers = 0
    while True:
        # both roles run their phase, then meet at the barrier; the consumers first, so that the buffer they release on leaving
        # the previous barrier is free -- and the one they now read is not -- when the producers' checks run
        b = next(co, None)
        a = next(pr, None)
        if a is None and b is None:
            break
        assert a == b == "barrier", "the roles execute different numbers of barriers: the kernel would hang (%s / %s)" % (a, b)
        barriers += 1
    return log, barriers


@pytest.mark.parametrize("nchunk", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("ntile", [1, 2, 3, 7, 30])
@pytest.mark.parametrize("tiles_per_item", [1, 2, 5, 1000])
def test_every_chunk_is_staged_once_consumed_once_and_the_barriers_match(nchunk, ntile, tiles_per_item):
    tiles_b = [3 + i // tiles_per_item for i in range(ntile)]  # batch items of a contiguous tile range: non-decreasing, steps of 1
    log, barriers = run_workgroup(tiles_b, nchunk)
    total = ntile * nchunk
    assert barriers == 2 + ntile * (nchunk + 1)
    assert [g for kind, g, _ in log if kind == "stage"] == list(range(total))
    assert [g for kind, g, _ in log if kind == "consume"] == list(range(total))
    staged_in = {g: ph for kind, g, ph in log if kind == "stage"}
    consumed_in = {g: ph for kind, g, ph in log if kind == "consume"}
    for g in range(total):
        assert staged_in[g] < consumed_in[g]
        assert consumed_in[g] - staged_in[g] <= 3  # the producers run at most two chunks (+ one epilogue phase) ahead
    # steady state: exactly one phase per tile without staging (the first MFMA phase after an epilogue phase)
    if ntile >= 4:  # (away from the first tile, which is staged from scratch, and the last two, where the chunks run out)
        phases_with_stage = {ph for ph in staged_in.values() if ph >= 0}
        idle = [ph for ph in range(nchunk + 1, (ntile - 2) * (nchunk + 1)) if ph not in phases_with_stage]
        assert all(ph % (nchunk + 1) == 0 for ph in idle) and len(idle) == ntile - 3


def test_the_model_catches_a_broken_schedule():
    """The checks are live: producers that run three chunks ahead with two buffers overwrite a chunk that is still being read."""
    lds = {"buf": [{"state": "free", "chunk": None, "phase": None} for _ in range(2)],
           "table": [{"b": None, "written_in_phase": None} for _ in range(2)], "table_in_use": {}, "reading": None}
    log = []
    pr, co = producers([0] * 6, 2, lds, log, depth=3), consumers([0] * 6, 2, lds, log)
    with pytest.raises(AssertionError, match="still holds chunk"):
        for _ in itertools.count():
            b = next(co, None)
            a = next(pr, None)
            if a is None and b is None:
                break


def f16s_tile_of_workgroup(bx, nx, row):
    """Mirror of the blockIdx.x -> column tile mapping of conv1d_f16s_kernel (st2_conv1d_f16s_impl.h): `row` = blockIdx.y +
    gridDim.y * blockIdx.z, workgroups are dealt to the 8 XCDs round-robin by their linear id."""
    o = (nx * row) & 7
    t = bx + o
    xcd = t & 7
    before, first_mine = 0, 0
    for r in range(8):
        t_first = o + ((r - o) & 7)
        cnt = (nx + o - 1 - t_first) // 8 + 1 if t_first < nx + o else 0
        if r < xcd:
            before += cnt
        if r == xcd:
            first_mine = t_first
    return before + (t - first_mine) // 8, xcd


@pytest.mark.parametrize("nx", [1, 2, 7, 8, 9, 63, 188, 469, 938])
def test_f16s_xcd_tile_order_is_a_bijection_with_contiguous_runs(nx):
    """Every column tile of a grid row is computed exactly once whatever the row's offset in the dispatch order, and the tiles one
    XCD receives form ONE contiguous run (so that neighbouring tiles, which share the tap halo's cache lines, share an L2)."""
    for row in range(8):
        tiles = [f16s_tile_of_workgroup(bx, nx, row) for bx in range(nx)]
        assert sorted(t for t, _ in tiles) == list(range(nx))
        by_xcd = {}
        for t, x in tiles:
            by_xcd.setdefault(x, []).append(t)
        for x, ts in by_xcd.items():
            ts.sort()
            assert ts == list(range(ts[0], ts[0] + len(ts))), (nx, row, x)
        for bx, (t, x) in enumerate(tiles):  # the XCD the mapping assumes is the one the round-robin dispatch gives the workgroup
            assert x == (bx + nx * row) % 8
