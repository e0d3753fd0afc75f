"""Does an HBM-bound activation pass hide under an MFMA-bound conv of ANOTHER tensor when the two are queued on two
streams?  (The question behind the interleaved decoder schedule of round 4 -- profiles/r04/r04y_interleaved_schedule.patch --
which lost 98 vs 71 ms single-stream in visit r04y.  Answer, profiles/r04/r04z2_overlap_probe.log: no -- each side slows
down by the other's share, the wall time equals back-to-back execution.)

For a resblock shape: N convs (st2_conv1d_xs, residual epilogue) on stream A, alone; K activation passes (st2_act_split,
AdaIN + Snake) on stream B, alone; then both at once, K chosen so that the two spans are about equal.  Printed per
variant of stream B: the span of each stream, the slow-down of each kind against running
alone, and the time the same work takes back to back.
"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from styletts2_amd import _lib, ops, pipeline, weights

dev = torch.device("cuda")
lib = _lib.load()
N = int(os.environ.get("OV_CONVS", "8"))


def span(fn_a, fn_b, sa, sb):
    """(ms on A, ms on B, ms wall) of fn_a queued on sa and fn_b on sb (either may be None)."""
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    cur = torch.cuda.current_stream()
    ev[0].record(cur)
    for st, fn, e0, e1 in ((sa, fn_a, ev[1], ev[2]), (sb, fn_b, ev[3], ev[4])):
        if fn is None:
            continue
        st.wait_event(ev[0])
        with torch.cuda.stream(st):
            e0.record(st)
            fn()
            e1.record(st)
    torch.cuda.synchronize()
    a = ev[1].elapsed_time(ev[2]) if fn_a else 0.0
    b = ev[3].elapsed_time(ev[4]) if fn_b else 0.0
    ends = [ev[0].elapsed_time(e) for e, f in ((ev[2], fn_a), (ev[4], fn_b)) if f]
    return a, b, max(ends)


def shape(Cc, L, ks, B=32):
    g = torch.Generator(device=dev).manual_seed(0)
    pitch = (L + 31) // 32 * 32
    x1 = torch.randn(B, Cc, pitch, device=dev, generator=g)[:, :, :L]
    x2 = torch.randn(B, Cc, pitch, device=dev, generator=g)[:, :, :L]
    out = torch.empty((B, Cc, pitch), device=dev)[:, :, :L]
    w = torch.randn(Cc, Cc, ks, device=dev, generator=g) / math.sqrt(Cc * ks)
    wt = weights.pack_conv_f16s(w).to(dev)
    bias = torch.randn(Cc, device=dev, generator=g)
    h = torch.randn(B, 2 * Cc, device=dev, generator=g) * 0.3
    alpha = torch.rand(Cc, device=dev, generator=g) + 0.5
    st = ops.instnorm_stats(x2)
    kw = dict(pro=ops.PRO_ADAIN_SNAKE, stats=st, gamma=h[:, :Cc], beta=h[:, Cc:], alpha=alpha)
    xs1 = ops.activate(x1, **kw)
    # the activation pass writes into planes of its own (allocated once: no allocator traffic inside the spans)
    xs2 = ops.activate(x2, **kw)
    cg, Lp = xs2.cg, xs2.Lp
    gbs = h.stride(0)

    def conv_n(n):
        def f():
            for _ in range(n):
                ops.conv1d_xs(xs1, wt, Cc, ks, pad_left=(ks - 1) // 2, bias=bias, out=out, res=x1)
        return f

    def act_n(n):
        def f():
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(n):
                _lib.check(lib.st2_act_split(x2.data_ptr(), x2.stride(0), x2.stride(1), B, Cc, L, ops.PRO_ADAIN_SNAKE, 0.0,
                                             st.data_ptr(), kw["gamma"].data_ptr(), kw["beta"].data_ptr(), gbs, 0, 1,
                                             alpha.data_ptr(), ops.x_scale_for(ops.PRO_ADAIN_SNAKE), xs2.data.data_ptr(),
                                             cg, Lp, ops.XS_HALO, s), "st2_act_split")
        return f
    return conv_n, act_n


def main():
    sa = torch.cuda.Stream(dev)
    streams = {"plain": torch.cuda.Stream(dev), "high-priority": torch.cuda.Stream(dev, priority=-1)}
    keep = []
    for cus in (32, 64, 128):
        try:
            ps = pipeline.PartitionedStreams(dev, cus)
            keep.append(ps)
            streams["%d-CU mask" % cus] = ps.front
        except Exception as e:
            print("no CU-masked stream:", e)
    for Cc, L, ks in ((128, 48001, 11), (256, 8000, 7)):
        conv_n, act_n = shape(Cc, L, ks)
        conv_n(2)(); act_n(2)()
        c_alone = span(conv_n(N), None, sa, None)[0] / N
        print("shape C=%d L=%d k=%d: conv alone %.4f ms" % (Cc, L, ks, c_alone), flush=True)
        act_n(2)()
        a_alone = span(None, act_n(N), None, streams["plain"])[1] / N
        K = max(1, int(round(N * c_alone / a_alone)))
        for sname, sb in streams.items():
            a, b, wall = span(conv_n(N), act_n(K), sa, sb)
            seq = N * c_alone + K * a_alone
            print("  act alone %.4f ms | B=%-13s %d convs ‖ %3d acts: conv x%.2f  act x%.2f  wall %.2f ms vs %.2f back to back "
                  "(%+.1f %%)" % (a_alone, sname, N, K, a / (N * c_alone), b / (K * a_alone), wall, seq,
                                  100.0 * (wall - seq) / seq), flush=True)


if __name__ == "__main__":
    main()
