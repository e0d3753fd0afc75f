#!/usr/bin/env python
"""One screen of a bench.py JSON line: ms/step, schedules, per-class conv times (timed region | un-overlapped), what the
autotuner chose, the box probe's headline numbers.  `python tools/bench_summary.py FILE [FILE ...]`"""
import json
import sys


def _short(name):
    return name.replace(" tiles, 3 wg/CU, dispatch order", "").replace(" tiles, 2 wg/CU", "").replace(
        "XCD-aware tile order", "xcd").replace("16-channel chunks", "c16").replace(" + ", "+")


def main():
    for path in sys.argv[1:]:
        try:
            lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
            d = json.loads(lines[-1])
        except Exception as e:
            print("%s: unreadable (%r)" % (path, e))
            continue
        cfg, roof = d.get("config", {}), d.get("roofline", {})
        print("%s: %.2f ms/step = %.0f %s, schedule %s %s, frac %.3f, conv %.1f ms/step" % (
            path, d["ms_per_step"], d["value"], d["unit"], cfg.get("schedule"), cfg.get("schedules_ms_per_step"),
            roof.get("frac") or 0.0, roof.get("conv_ms_per_step_all_classes") or 0.0))
        cl = roof.get("ceiling_live")
        if cl:
            print("   live ceiling: %.0f TFLOP/s f16 on random operands @ %.2f GHz = %.3f of the roof; dominant class at %.2f of it" % (
                cl["mfma_tflops_random_operands"], cl.get("clock_ghz") or 0.0, cl["frac_of_peak"], cl["kernel_frac_of_it"]))
        un = {(c["ks"], c["C_in"], c["C_out"], c["L"]): c for c in (roof.get("unoverlapped") or {}).get("classes", [])}
        for c in roof.get("classes", []):
            u = un.get((c["ks"], c["C_in"], c["C_out"], c["L"]))
            print("   k%-2d C%4d->%4d L%6d  %.4f ms  frac %.3f | alone %s" % (
                c["ks"], c["C_in"], c["C_out"], c["L"], c["avg_launch_ms"], c["frac"],
                ("%.4f ms frac %.3f" % (u["avg_launch_ms"], u["frac"])) if u else "-"))
        tune = cfg.get("conv_autotune")
        if isinstance(tune, list):
            for r in tune:
                if len(r["ms"]) > 1:
                    print("   tune %-28s -> %-40s %s" % (r["class"], r["chosen"], {_short(k): round(v, 4) for k, v in r["ms"].items()}))
        box = d.get("box") or {}
        pr = box.get("probe") or {}
        if pr and "error" not in pr:
            print("   box: %s CUs, mfma %s, sets %s, hbm %.0f GB/s, census %s" % (
                pr.get("cus"), {k: (round(v["tflops"]), round(v["clock_ghz"], 2)) for k, v in pr.get("mfma", {}).items()},
                [(s["mb"], round(s["chase_ns_median"]), round(s["stream2_gbps"])) for s in pr.get("sets", [])],
                pr.get("hbm_copy", {}).get("gbps_read_plus_write", 0.0),
                {k: pr["census"][k] for k in ("cus_seen", "wg_per_cu_min", "wg_per_cu_max", "wg_not_on_xcd_id_mod_8",
                                              "rounds_of_30us") if k in pr.get("census", {})}))
            for k in ("scatter_store", "row_store"):
                if k in pr:
                    print("   %s: median %s cycles / workgroup, max %s, slow CUs %s %s" % (
                        k, pr[k].get("cycles_per_wg_median"), pr[k].get("cycles_per_wg_max"), pr[k].get("n_slow_cus"),
                        pr[k].get("slow_cus")[:8]))
            ch = box.get("cu_health")
            if ch:
                print("   cu_health: %s slow CUs %s, XCD ends %s us; healthy-CU streams %s" % (
                    ch.get("n_slow_cus"), [(c["xcc"], c["se"], c["cu"], c["x_median"]) for c in ch.get("slow_cus", [])][:8],
                    ch.get("xcd_end_us"), box.get("healthy_cu_streams")))
            sf = box.get("sysfs", {})
            print("   sysfs: %s" % {k: sf[k] for k in ("current_compute_partition", "current_memory_partition", "power1_cap",
                                                         "vbios_version") if k in sf})
            print("   sensors: %s" % box.get("sensors_during_calibration"))
        cb = d.get("cpu_baseline")
        if cb:
            print("   cpu_baseline: %.1f %s (%s, %s cores)" % (cb["value"], cb["unit"], cb["kind"], cb["cores"]))


if __name__ == "__main__":
    main()
