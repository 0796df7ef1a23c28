#!/usr/bin/env python
"""Recompute the dominant kernel's roofline fraction from the profile artefacts alone and cross-check it.

    python tools/roofline_check.py <bench line of the profiled command (json)> <kernel trace summary> <pmc_traffic.json> <pmc sq summary>

Inputs: the JSON line the profiled `bench.py --steps K --warmup W` printed (mean real E of the timed steps:
config.edges_per_traj_mean, lb_edge_accounting), tools/rocpd_summary.py's table of the same run (avg_real_us of the edge
kernel), the FETCH_SIZE / WRITE_SIZE table and the SQ_INSTS_* table.  Checks:
  1. algorithmic bytes (E_mean * 1032 + B*N * 1024) / rocprofv3's average launch time  vs  the bench line's roofline.frac
     (HIP events inside bench.py): must agree within 3 %;
  2. SQ_INSTS_MFMA per launch / (E_mean / 16 tiles)  vs  192 MFMAs per 16-edge tile in the ISA: within 2 % - the edge
     count the bytes are computed from is the one the kernel actually walked;
  3. PMC traffic / algorithmic bytes (waste when well above 1).
"""
import json
import re
import sys

D = 128


def main():
    bench = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    cfg, roof = bench["config"], bench["roofline"]
    B, N = cfg["batch_per_gpu"], cfg["n_particles"]
    E = cfg["edges_per_traj_mean"] * B
    byts = E * (2 * D * 4 + 8) + B * N * 2 * D * 4
    kern = roof["kernel"].split("<")[0].split(" ")[0]
    us = None
    for l in open(sys.argv[2]):
        if kern + "<" in l and "false" in l:   # the storing variant (SKIP = false is the first template argument)
            m = re.match(r"(void )?" + kern + r"<(\w+),", l.strip())
            if m and m.group(2) == "false":
                us = float(l.split()[-1])
                break
    print(f"edge kernel {kern}: E_mean {E:.0f} ({E / B:.1f} per trajectory), {E / 16:.0f} tiles, algorithmic {byts / 1e6:.1f} MB per launch")
    print(f"bench line (HIP events): {roof['us_per_launch']:.2f} us per launch -> frac {roof['frac']:.4f}")
    ok = True
    if us:
        frac = byts / (us * 1e-6) / 8e12
        dev = frac / roof["frac"] - 1
        print(f"rocprofv3 kernel trace : {us:.2f} us per launch -> frac {frac:.4f}  ({100 * dev:+.1f} % vs the bench line) "
              f"{'OK' if abs(dev) <= 0.03 else 'MISMATCH'}")
        ok &= abs(dev) <= 0.03
    try:
        tab = json.load(open(sys.argv[3]))
        for key, ent in tab.items():
            for k, v in ent.items():
                if k.startswith(("void " + kern + "<false", kern + "<false")):
                    print(f"PMC traffic            : {v['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch = "
                          f"{v['hbm_bytes_per_launch'] / byts:.3f} x algorithmic")
    except Exception as exc:
        print("PMC traffic: n/a", exc)
    try:
        for l in open(sys.argv[4]):
            if ("void " + kern + "<false") in l and "SQ_INSTS_MFMA=" in l:
                mf = float(re.search(r"SQ_INSTS_MFMA=([0-9.e+]+)", l).group(1))
                per_tile = mf / (E / 16)
                print(f"SQ_INSTS_MFMA          : {mf:.4g} per launch / {E / 16:.0f} tiles = {per_tile:.1f} per tile (ISA: 192) "
                      f"{'OK' if abs(per_tile / 192 - 1) <= 0.02 else 'MISMATCH'}")
                ok &= abs(per_tile / 192 - 1) <= 0.02
                for c in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU"):
                    m = re.search(c + r"=([0-9.e+]+)", l)
                    if m:
                        print(f"{c:23s}: {float(m.group(1)) / (E / 16):.1f} per tile")
            if ("void " + kern + "<false") in l and "SQ_INSTS_VALU=" in l:
                v = float(re.search(r"SQ_INSTS_VALU=([0-9.e+]+)", l).group(1))
                print(f"SQ_INSTS_VALU          : {v / (E / 16):.1f} per tile")
                m = re.search(r"SQ_VALU_MFMA_BUSY_CYCLES=([0-9.e+]+)", l)
                w = re.search(r"SQ_WAVE_CYCLES=([0-9.e+]+)", l)
                g = re.search(r"GRBM_GUI_ACTIVE=([0-9.e+]+)", l)
                if m and g:
                    print(f"matrix pipe busy       : {float(m.group(1)) / 1024 / (float(g.group(1)) / 8):.3f} of the launch (per SIMD)")
    except Exception as exc:
        print("SQ counters: n/a", exc)
    print("self-check", "PASSED" if ok else "FAILED")


if __name__ == "__main__":
    main()
