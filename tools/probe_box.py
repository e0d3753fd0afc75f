#!/usr/bin/env python
"""FIRST command of every GPU visit: which class of box is this?

    python tools/probe_box.py [--out gpurun_out/box.json] [--level 1] [--kit]

Prints the box fingerprint (benchdata/boxinfo.py: device properties, sysfs, the library's micro-probe) and times the
launch class that separates the box classes -- st2_conv1d_xs k = 7, C = 256 -> 256, L = 8 000, B = 32, dilation 1 (the
first stage of Generator.forward, Modules/istftnet.py:358-375) -- in every bitwise-equivalent build.  Builder-class boxes run
the rule's build (128 x 256 tiles) in ~0.70 ms, driver-class boxes in ~1.22 ms.  Exit code 0 always; the
last line is `BOX_CLASS fast|slow <ms of the rule's build>` so that a visit script can branch into the full kit
(`--kit` prints the commands: rocprofv3 --pmc passes over this launch, xs_bench ablations, the per-workgroup timeline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

SLOW_MS = 0.9
CLASSES = [(7, 256, 8000, 1), (11, 256, 8000, 1), (3, 256, 8000, 1), (11, 128, 48001, 1), (7, 128, 48001, 3),
           (3, 1024, 400, 1)]


def time_class(ks, C, L, dil, B=32, variants=(0, 1, 2, 3), reps=5):
    """ms / launch of one shape class per variant, through the library's own entry point (st2_act_split + st2_conv1d_xs)."""
    from styletts2_amd import ops, weights
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(ks * 1000 + C)
    w = torch.randn(C, C, ks, generator=g) * (1.0 / (C * ks) ** 0.5)
    wt = weights.pack_conv_f16s(w).to(dev)
    pitch = (L + 31) // 32 * 32
    x = torch.randn(B, C, pitch, generator=g).to(dev)[:, :, :L]
    res = torch.randn(B, C, pitch, generator=g).to(dev)[:, :, :L]
    out = torch.empty(B, C, pitch, device=dev)[:, :, :L]
    bias = torch.randn(C, generator=g).to(dev)
    xs = ops.activate(x)
    pad = (ks - 1) * dil // 2
    res_ms = {}
    for v in variants:
        if v & 1 and ks < 7:
            continue
        ops.conv_tune_set(ks, C, C, L, B, v)
        ops.conv1d_xs(xs, wt, C, ks, dil=dil, pad_left=pad, bias=bias, res=res, out=out, want_stats=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(2):
            e0.record()
            for _ in range(reps):
                ops.conv1d_xs(xs, wt, C, ks, dil=dil, pad_left=pad, bias=bias, res=res, out=out, want_stats=True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        res_ms[ops.tune_variant_name(v)] = round(best, 4)
    ops.conv_tune_set(ks, C, C, L, B, -1)
    return res_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--quick", action="store_true", help="only the discriminating class")
    ap.add_argument("--no-health", action="store_true", help="skip st2_probe_cu_health (CU-masked streams do not survive "
                                                             "rocprofv3 --pmc: the slow kit's counter passes use this)")
    ap.add_argument("--kit", action="store_true", help="print the full-kit commands for a slow box")
    a = ap.parse_args()
    from benchdata import boxinfo
    t0 = time.time()
    fp = boxinfo.fingerprint(0, probe=True, level=a.level, health=not a.no_health)
    fp["conv_classes"] = []
    for ks, C, L, dil in (CLASSES[:1] if a.quick else CLASSES):
        variants = (0, 1, 2, 3) if ks >= 7 else (0, 2)
        with boxinfo.Sampler(0) as smp:
            ms = time_class(ks, C, L, dil, variants=variants)
        flop = 2.0 * 32 * C * C * ks * L
        fp["conv_classes"].append({"ks": ks, "C": C, "L": L, "dil": dil, "ms": ms,
                                   "frac_best": round(flop / (min(ms.values()) * 1e-3) / 1e12 / (2500.0 / 3), 4),
                                   "sensors": smp.summary()})
    fp["wall_s"] = round(time.time() - t0, 1)
    text = json.dumps(fp, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(text + "\n")
    rule_ms = fp["conv_classes"][0]["ms"].get("128x256 tiles, 2 wg/CU", 0.0)
    n_slow = (fp.get("cu_health") or {}).get("n_slow_cus", 0) or 0
    cls = "slow" if (rule_ms > SLOW_MS or n_slow > 0) else "fast"
    print("cu_health: %s slow CUs %s, XCD ends %s us" % (n_slow, [(c["xcc"], c["se"], c["cu"], c["x_median"]) for c in
                                                                  (fp.get("cu_health") or {}).get("slow_cus", [])][:8],
                                                          (fp.get("cu_health") or {}).get("xcd_end_us")), file=sys.stderr)
    if a.kit or cls == "slow":
        print("# full kit for this box (run inside the same visit):", file=sys.stderr)
        print("#   bash tools/gpu_visit.sh slowkit", file=sys.stderr)
    print("BOX_CLASS %s %.4f" % (cls, rule_ms))


if __name__ == "__main__":
    main()
