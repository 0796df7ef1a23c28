#!/usr/bin/env python
"""Shader clock and package power of the GPU while something runs on it ("power bound" as a tracked
metric, not two rocm-smi samples).

The amdgpu hwmon node of a card (/sys/class/drm/cardN/device/hwmon/hwmonM) gives `freq1_input` (shader clock, Hz),
`power1_input` (package power, microwatt) and `power1_cap`; reading them costs a few microseconds, so a thread can sample at
50 - 100 Hz without disturbing the launching thread.  Two uses:

  * library:  with PowerProbe(hz=100) as p: <run the workload>;  p.summary(n_units) -> mean / min / max sclk, mean power,
    joules per unit - bench.py's `power` object and `roofline.limiter`;
  * CLI:      python tools/power_probe.py [--hz 20] [--skip 2.0] [--label txt] -- <command ...>
    samples while the command runs (after `--skip` seconds of warm-up) and prints ONE summary line - tools/energy_ab.sh.

The card is the one whose PCI address matches HIP device 0 (torch) when that can be told, else the busiest one (highest mean
power over the window).
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
import threading
import time


def _cards():
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(hw, "power1_input")) and os.path.exists(os.path.join(hw, "freq1_input")):
            dev = os.path.realpath(os.path.join(hw, "..", ".."))
            out.append((hw, os.path.basename(dev)))   # (hwmon dir, PCI address 0000:bb:dd.f)
    return out


def _read(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return float("nan")


class PowerProbe:
    def __init__(self, hz: float = 50.0, pci: str | None = None):
        self.dt = 1.0 / hz
        self.cards = _cards()
        self.pci = pci.lower() if pci else None
        self.samples = {hw: [] for hw, _ in self.cards}   # hw -> [(t, sclk_hz, power_uw)]
        self._stop = threading.Event()
        self._th = None

    def __enter__(self):
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th:
            self._th.join()

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            for hw, _ in self.cards:
                self.samples[hw].append((t, _read(os.path.join(hw, "freq1_input")), _read(os.path.join(hw, "power1_input"))))
            rest = self.dt - (time.perf_counter() - t)
            if rest > 0:
                self._stop.wait(rest)

    def _pick(self):
        if not self.cards:
            return None
        if self.pci:
            for hw, addr in self.cards:
                if addr.lower().endswith(self.pci) or self.pci.endswith(addr.lower()):
                    return hw
        best, best_p = None, -1.0
        for hw, _ in self.cards:
            ps = [p for _, _, p in self.samples[hw] if p == p]
            m = sum(ps) / len(ps) if ps else -1.0
            if m > best_p:
                best, best_p = hw, m
        return best

    def summary(self, units: float | None = None, t0: float | None = None, t1: float | None = None):
        """Mean shader clock / power over the samples in [t0, t1] (perf_counter stamps; default: all).  `units`: how many
        work units (steps, launches) ran in that window -> joules per unit."""
        hw = self._pick()
        if hw is None:
            return None
        rows = [(t, f, p) for t, f, p in self.samples[hw] if f == f and p == p and (t0 is None or t >= t0) and (t1 is None or t <= t1)]
        if not rows:
            return None
        f = [r[1] / 1e6 for r in rows]
        p = [r[2] / 1e6 for r in rows]
        span = rows[-1][0] - rows[0][0]
        cap = _read(os.path.join(hw, "power1_cap")) / 1e6
        res = {"samples": len(rows), "window_s": round(span, 3), "hz": round(len(rows) / span, 1) if span > 0 else None,
               "sclk_mhz_mean": round(sum(f) / len(f), 1), "sclk_mhz_min": round(min(f), 1), "sclk_mhz_max": round(max(f), 1),
               "power_w_mean": round(sum(p) / len(p), 1), "power_w_max": round(max(p), 1),
               "power_cap_w": round(cap, 1) if cap == cap else None,
               "source": hw + "/{freq1_input,power1_input}"}
        if units and span > 0:
            # energy of the window = mean power x the window the units ran in (caller passes matching t0 / t1)
            res["joules_per_unit"] = (sum(p) / len(p)) * ((t1 - t0) if (t0 is not None and t1 is not None) else span) / units
        return res


def hip_pci_address():
    """PCI address of HIP device 0 as sysfs spells it, or None."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(0)
        dom, bus, dev = getattr(pr, "pci_domain_id", None), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)
        if bus is None:
            return None
        return f"{dom or 0:04x}:{bus:02x}:{dev or 0:02x}.0"
    except Exception:
        return None


def main():
    argv = sys.argv[1:]
    hz, skip, label = 20.0, 2.0, None
    while argv and argv[0] != "--":
        if argv[0] == "--hz":
            hz = float(argv[1]); argv = argv[2:]
        elif argv[0] == "--skip":
            skip = float(argv[1]); argv = argv[2:]
        elif argv[0] == "--label":
            label = argv[1]; argv = argv[2:]
        else:
            raise SystemExit(__doc__)
    cmd = argv[1:]
    if not cmd:
        raise SystemExit(__doc__)
    with PowerProbe(hz=hz) as p:
        t_begin = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_end = time.perf_counter()
    s = p.summary(t0=t_begin + skip, t1=t_end)
    tail = [ln for ln in r.stdout.strip().splitlines() if ln.strip()][-3:]
    print(f"{label or ' '.join(cmd)[:60]:60s} | " + (
        f"n={s['samples']:4d} {s['hz']} Hz  sclk {s['sclk_mhz_mean']:7.1f} MHz ({s['sclk_mhz_min']:.0f}..{s['sclk_mhz_max']:.0f})  "
        f"power {s['power_w_mean']:7.1f} W (max {s['power_w_max']:.0f}, cap {s['power_cap_w']})" if s else "no hwmon samples"))
    for ln in tail:
        print("    " + ln[:200])


if __name__ == "__main__":
    main()
