"""What box is this?  Fingerprint of the GPU a bench / probe runs on, for runs on boxes nobody can log into.

`fingerprint(dev_index)` = torch's device properties + whatever the amdgpu driver exposes under sysfs for that card
(compute / memory partition, DPM clock tables, power cap, firmware versions) + the library's micro-probe
(`st2_probe_box`: matrix-pipe clock, cache-level latencies, weight-stream and HBM bandwidth, workgroup census).  Everything
is best effort: a file that does not exist on this kernel is simply absent from the result.

`Sampler` reads the card's live sensors (shader clock, power, temperature) from a background thread while something
else runs -- the bench uses it around its calibration steps so that the JSON line says what clock and power the chip
sustained under the workload itself.
"""
import glob
import os
import threading
import time

_SYSFS_FILES = ["current_compute_partition", "current_memory_partition", "available_compute_partition",
                "pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "power_dpm_force_performance_level",
                "vbios_version", "mem_info_vram_total", "mem_info_vram_used", "gpu_busy_percent", "pcie_bw",
                "current_link_speed", "current_link_width", "xcp_config", "numa_node", "device", "revision",
                "subsystem_device"]
_HWMON_FILES = ["power1_cap", "power1_cap_max", "power1_cap_default", "power1_average", "power1_input", "freq1_input",
                "freq2_input", "temp1_input", "temp2_input", "temp3_input", "in0_input"]


def _read(path, limit=400):
    try:
        with open(path) as f:
            return f.read(limit).strip()
    except OSError:
        return None


def _cards():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(d, "vendor")) == "0x1002":
            out.append(d)
    return out


def _card_for(dev_index):
    """sysfs directory of the card behind torch device `dev_index` (matched by PCI address where torch exposes it; the only
    amdgpu card otherwise)."""
    cards = _cards()
    if not cards:
        return None
    try:
        import torch
        p = torch.cuda.get_device_properties(dev_index)
        want = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        for d in cards:
            if want in os.path.realpath(d):
                return d
    except Exception:
        pass
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if len(cards) == 1:
        return cards[0]
    try:
        return cards[int(vis.split(",")[dev_index])]
    except Exception:
        return cards[min(dev_index, len(cards) - 1)]


def sysfs(dev_index=0):
    d = _card_for(dev_index)
    if d is None:
        return {"note": "no amdgpu card under /sys/class/drm"}
    out = {"card": os.path.realpath(d)[-40:], "cards_on_host": len(_cards())}
    for name in _SYSFS_FILES:
        v = _read(os.path.join(d, name))
        if v is not None:
            out[name] = v
    for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        for name in _HWMON_FILES:
            v = _read(os.path.join(hw, name))
            if v is not None:
                out[name] = v
    fw = {}
    for f in sorted(glob.glob(os.path.join(d, "fw_version", "*"))):
        v = _read(f)
        if v is not None and v not in ("0x00000000",):
            fw[os.path.basename(f).replace("_fw_version", "")] = v
    if fw:
        out["fw"] = fw
    return out


def torch_props(dev_index=0):
    import torch
    p = torch.cuda.get_device_properties(dev_index)
    out = {}
    for k in ("name", "gcnArchName", "multi_processor_count", "total_memory", "L2_cache_size", "clock_rate",
              "memory_clock_rate", "memory_bus_width", "max_threads_per_multi_processor", "warp_size",
              "shared_memory_per_multiprocessor", "regs_per_multiprocessor", "pci_bus_id", "pci_device_id"):
        v = getattr(p, k, None)
        if v is not None:
            out[k] = v if isinstance(v, (int, float, str)) else str(v)
    return out


def host():
    out = {"cpus": os.cpu_count(), "kernel": os.uname().release}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                out["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_MAX_HW_QUEUES", "HSA_ENABLE_SDMA", "HSA_XNACK"):
        if k in os.environ:
            out[k] = os.environ[k]
    return out


def fingerprint(dev_index=0, probe=True, level=0, health=True):
    out = {"torch": torch_props(dev_index), "sysfs": sysfs(dev_index), "host": host()}
    if probe:
        try:
            from styletts2_amd import ops
            t = time.time()
            out["probe"] = ops.probe_box(level)
            out["probe"]["wall_s"] = round(time.time() - t, 2)
        except Exception as e:  # a diagnostic must never be in the way of the measurement
            out["probe"] = {"error": repr(e)}
        try:
            if not health:
                raise RuntimeError("skipped on request")
            from styletts2_amd import ops
            rep, mask, n = ops.probe_cu_health()
            rep["healthy_mask"] = ["0x%08x" % w for w in mask]
            out["cu_health"] = rep
        except Exception as e:
            out["cu_health"] = {"error": repr(e)}
    return out


class Sampler:
    """Background reader of the card's live sensors: `with Sampler(dev) as s: work()`, then `s.summary()` = min / mean / max
    of the shader clock (MHz), socket power (W) and temperatures seen while `work()` ran."""

    def __init__(self, dev_index=0, period_s=0.02):
        self.period = period_s
        self.rows = []
        self._stop = threading.Event()
        self._files = {}
        d = _card_for(dev_index)
        if d:
            for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                for name in ("freq1_input", "power1_average", "power1_input", "temp1_input", "temp2_input"):
                    p = os.path.join(hw, name)
                    if os.path.exists(p):
                        self._files[name] = p
            p = os.path.join(d, "gpu_busy_percent")
            if os.path.exists(p):
                self._files["gpu_busy_percent"] = p
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            row = {}
            for k, p in self._files.items():
                v = _read(p, 32)
                try:
                    row[k] = float(v)
                except (TypeError, ValueError):
                    pass
            if row:
                self.rows.append(row)
            self._stop.wait(self.period)

    def __enter__(self):
        if self._files:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(1.0)
        return False

    def summary(self):
        if not self.rows:
            return {"note": "no live sensors readable"}
        scale = {"freq1_input": 1e-6, "power1_average": 1e-6, "power1_input": 1e-6, "temp1_input": 1e-3, "temp2_input": 1e-3}
        unit = {"freq1_input": "sclk_mhz", "power1_average": "power_w", "power1_input": "power_w", "temp1_input": "temp1_c",
                "temp2_input": "temp2_c", "gpu_busy_percent": "busy_pct"}
        out = {"samples": len(self.rows)}
        for k in self._files:
            vals = [r[k] * scale.get(k, 1.0) for r in self.rows if k in r]
            if vals:
                out[unit.get(k, k)] = {"min": round(min(vals), 1), "mean": round(sum(vals) / len(vals), 1),
                                       "max": round(max(vals), 1)}
        return out
