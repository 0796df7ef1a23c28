#!/usr/bin/env python
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    python tools/pmc_traffic.py <key e.g. tgv3d_b8> <fetch results.db> <write results.db> [out.json]

Per kernel: average FETCH_SIZE / WRITE_SIZE per launch (KB, summed over the counter's instances,
no-op launches - duration < 20 % of the kernel's median, i.e. the poisoned launches after a
neighbor-list overflow during warm-up - excluded) and
    hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
with the x2 on FETCH_SIZE that MI355X_MICROARCH.md (section HBM) prescribes for gfx950's wide
coalesced reads.
"""
import json
import os
import sqlite3
import statistics
import sys
from collections import defaultdict


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    t = {r[0].split("_0000")[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}
    names = {r[0]: r[1] for r in c.execute(f"select id, display_name from {t['rocpd_info_kernel_symbol']}")}
    disp = list(c.execute(f"select kernel_id, start, end, event_id from {t['rocpd_kernel_dispatch']}"))
    pmc = {r[0]: r[1] for r in c.execute(f"select id, name from {t['rocpd_info_pmc']}")}
    val = defaultdict(float)
    for ev, pid, v in c.execute(f"select event_id, pmc_id, value from {t['rocpd_pmc_event']}"):
        if pmc[pid] == counter:
            val[ev] += v
    durs = defaultdict(list)
    for kid, s, e, ev in disp:
        durs[names[kid]].append((e - s, ev))
    out = {}
    for k, lst in durs.items():
        med = statistics.median(d for d, _ in lst)
        keep = [ev for d, ev in lst if d >= 0.2 * med]
        if keep:
            out[k] = sum(val[ev] for ev in keep) / len(keep)
    return out


def main():
    key, fdb, wdb = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    tab = json.load(open(path)) if os.path.exists(path) else {}
    ent = {}
    for k in f:
        if k.startswith(("void k_", "k_")):
            ent[k[:80]] = {"fetch_kb": round(f[k], 1), "write_kb": round(w.get(k, 0.0), 1),
                           "hbm_bytes_per_launch": int((2 * f[k] + w.get(k, 0.0)) * 1024)}
    tab[key] = ent
    json.dump(tab, open(out_path or path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(ent.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
        print(f"{k[:60]:60s} fetch {v['fetch_kb']:12.1f} KB  write {v['write_kb']:12.1f} KB  hbm {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
