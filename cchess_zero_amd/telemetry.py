"""Clock / power telemetry of the GPU a rank runs on, for bench.py's roofline block (measurement only).

Two independent sources:
  * `GpuSampler`: a background thread that reads the amdgpu hwmon files of the device (socket power `power1_input`, shader
    clock `freq1_input`, the power cap) every 50 ms while a timed region runs — the driver's own view, the numbers
    `rocm-smi --showpower --showclocks` prints.  The device is found by its PCI address (torch's device properties ->
    /sys/class/drm/card*/device); when sysfs is not readable the amdsmi Python binding is tried; when neither works the
    sampler reports {"source": None} and the bench line says so — it never raises into the timed code.
  * `ClockProbe`: the trunk kernels themselves (cz_set_clock_probe): every workgroup stamps the shader-clock cycle counter and
    the constant 100 MHz reference clock at its start and at the end of its last layer; cycles / time is the clock the
    kernel REALLY ran at under the power governor, per workgroup, with no sampling artefact.
"""
import glob
import os
import threading
import time


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


def hwmon_dir(device_index=0):
    """hwmon directory of torch device `device_index` (matched by PCI address), or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if os.path.basename(os.path.realpath(card)).lower() == bdf:
                hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*")))
                return hw[0] if hw else None
        except Exception:
            continue
    return None


class GpuSampler:
    def __init__(self, device_index=0, period=0.05):
        self.period = period
        self.samples = []          # (t, watts or None, sclk MHz or None)
        self._stop = threading.Event()
        self._thread = None
        self.source = None
        self.cap_w = None
        self._hw = hwmon_dir(device_index)
        self._smi = None
        if self._hw and _read_int(os.path.join(self._hw, "power1_input")) is not None:
            self.source = "sysfs hwmon (%s: power1_input, freq1_input)" % self._hw
            cap = _read_int(os.path.join(self._hw, "power1_cap"))
            self.cap_w = cap / 1e6 if cap else None
        else:
            try:
                import amdsmi
                amdsmi.amdsmi_init()
                hs = amdsmi.amdsmi_get_processor_handles()
                if hs:
                    self._smi = (amdsmi, hs[min(device_index, len(hs) - 1)])
                    self.source = "amdsmi (gpu_metrics: current_socket_power, current_gfxclks)"
            except Exception:
                self._smi = None

    def _read(self):
        if self._hw and self._smi is None:
            p = _read_int(os.path.join(self._hw, "power1_input"))
            f = _read_int(os.path.join(self._hw, "freq1_input"))
            return (p / 1e6 if p is not None else None, f / 1e6 if f is not None else None)
        if self._smi is not None:
            amdsmi, h = self._smi
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                clk = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
                pw = m.get("current_socket_power")
                return (float(pw) if isinstance(pw, (int, float)) else None, float(sum(clk)) / len(clk) if clk else None)
            except Exception:
                return (None, None)
        return (None, None)

    def _run(self):
        while not self._stop.is_set():
            w, f = self._read()
            self.samples.append((time.perf_counter(), w, f))
            self._stop.wait(self.period)

    def start(self):
        if self.source is None:
            return self
        self.samples = []
        self._stop.clear()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2.0)
            self._thread = None
        return self.summary()

    def summary(self, skip_seconds=0.25):
        """Mean / min / max over the samples, the first `skip_seconds` dropped (the clocks ramp after an idle gap)."""
        if self.source is None or not self.samples:
            return {"source": self.source, "samples": 0}
        t0 = self.samples[0][0]
        rows = [s for s in self.samples if s[0] - t0 >= skip_seconds] or self.samples
        ws = [s[1] for s in rows if s[1] is not None]
        fs = [s[2] for s in rows if s[2] is not None]
        out = {"source": self.source, "samples": len(rows), "period_s": self.period, "power_cap_W": self.cap_w}
        if ws:
            out.update(power_W=sum(ws) / len(ws), power_W_min=min(ws), power_W_max=max(ws))
        if fs:
            out.update(sclk_MHz=sum(fs) / len(fs), sclk_MHz_min=min(fs), sclk_MHz_max=max(fs))
        return out


class ClockProbe:
    """cz_set_clock_probe around chosen trunk launches; effective clock = shader cycles / reference time per workgroup."""

    def __init__(self, ctx, max_workgroups):
        import torch
        self.ctx = ctx
        self.n = int(max_workgroups)
        self.buf = torch.zeros((self.n, 4), dtype=torch.int64, device=ctx.device)
        self.readings = []

    def arm(self):
        import ctypes as C
        from ._lib import check, lib
        check(lib().cz_set_clock_probe(self.ctx.h, C.c_void_p(self.buf.data_ptr()), self.n), "cz_set_clock_probe")

    def disarm(self, collect=True):
        """Switch the probe off; with collect=True read the stamps of the last probed launch (synchronises)."""
        import torch
        from ._lib import check, lib
        grid = lib().cz_clock_probe_last_grid(self.ctx.h)
        check(lib().cz_set_clock_probe(self.ctx.h, None, 0), "cz_set_clock_probe")
        if not collect or grid <= 0:
            return None
        torch.cuda.synchronize()
        b = self.buf[:grid].cpu().numpy()
        cyc = (b[:, 1] - b[:, 0]).astype("float64")
        ref = (b[:, 3] - b[:, 2]).astype("float64") * 10e-9     # s_memrealtime: 100 MHz
        ok = (cyc > 0) & (ref > 0)
        if not ok.any():
            return None
        ghz = cyc[ok] / ref[ok] / 1e9
        r = {"workgroups": int(ok.sum()), "clock_GHz": float(ghz.mean()), "clock_GHz_min": float(ghz.min()), "clock_GHz_max": float(ghz.max()),
             "workgroup_life_us": float(ref[ok].mean() * 1e6), "cycles_per_workgroup": float(cyc[ok].mean()),
             "launch_span_us": float((b[ok, 3].max() - b[ok, 2].min()) * 10e-3)}
        self.readings.append(r)
        return r

    def mean(self):
        if not self.readings:
            return None
        k = lambda name: sum(r[name] for r in self.readings) / len(self.readings)
        return {"launches_probed": len(self.readings), "effective_clock_GHz": k("clock_GHz"),
                "effective_clock_GHz_min_workgroup": min(r["clock_GHz_min"] for r in self.readings),
                "effective_clock_GHz_max_workgroup": max(r["clock_GHz_max"] for r in self.readings),
                "workgroup_life_us": k("workgroup_life_us"), "cycles_per_workgroup": k("cycles_per_workgroup"),
                "launch_span_us": k("launch_span_us"),
                "method": "every workgroup of the trunk kernel stamps s_memtime (shader-clock cycles) and s_memrealtime (100 MHz) at its start and after its last layer (cz_set_clock_probe); mean over workgroups and probed launches"}
