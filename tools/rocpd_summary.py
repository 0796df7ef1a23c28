#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2, rocpd SQLite output) runs into the text tables committed under
profiles/.

    python tools/rocpd_summary.py <results.db> [...]  > profiles/rNN_xxx.txt

For every database: per-kernel dispatch statistics (calls, total, average, min, max - the same
numbers `--stats` prints) and, when the run collected PMC counters, the per-launch average of each
counter per kernel (summed over the counter's hardware instances).
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\.kd$", "", name)
    m = re.match(r"_Z\d+(k_[a-z_0-9]+?)(I.*)?$", name)
    return name if not m else name


def tables(c):
    return {r[0].split("_0000")[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}


def main():
    for db in sys.argv[1:]:
        c = sqlite3.connect(db)
        t = tables(c)
        names = {r[0]: r[1] for r in c.execute(f"select id, display_name from {t['rocpd_info_kernel_symbol']}")}
        disp = list(c.execute(f"select kernel_id, start, end, event_id from {t['rocpd_kernel_dispatch']}"))
        stat = defaultdict(list)
        ev2k, ev2d = {}, {}
        for kid, s, e, ev in disp:
            stat[names.get(kid, str(kid))].append(e - s)
            ev2k[ev] = names.get(kid, str(kid))
            ev2d[ev] = e - s
        tot = sum(sum(v) for v in stat.values()) or 1
        print(f"== {db}")
        print("(avg_real_us / real: launches of at least 20 % of the kernel's median duration, i.e. without the "
              "no-op launches that follow a neighbor-list overflow during warm-up)")
        print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
              f"{'real':>6s} {'avg_real_us':>11s}")
        for k, v in sorted(stat.items(), key=lambda kv: -sum(kv[1])):
            med = sorted(v)[len(v) // 2]
            real = [x for x in v if x >= 0.2 * med]
            print(f"{k[:70]:70s} {len(v):7d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:9.2f} "
                  f"{max(v)/1e3:9.2f} {100*sum(v)/tot:6.2f} {len(real):6d} {sum(real)/len(real)/1e3:11.2f}")
        pmc_names = {r[0]: r[1] for r in c.execute(f"select id, name from {t['rocpd_info_pmc']}")}
        if pmc_names:
            acc = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(set)
            med = {k: sorted(v)[len(v) // 2] for k, v in stat.items()}
            for ev, pid, val in c.execute(f"select event_id, pmc_id, value from {t['rocpd_pmc_event']}"):
                k = ev2k.get(ev)
                if k is None or ev2d[ev] < 0.2 * med[k]:   # (round 5: REAL launches only, like avg_real_us)
                    continue
                acc[k][pmc_names[pid]] += val
                cnt[k].add(ev)
            cols = sorted(set(pmc_names.values()))
            print(f"-- PMC, average per REAL launch (summed over instances): {', '.join(cols)}")
            for k in sorted(acc, key=lambda kk: -sum(stat[kk])):
                n = len(cnt[k])
                vals = "  ".join(f"{cname}={acc[k][cname]/n:.4g}" for cname in cols)
                print(f"{k[:70]:70s} launches={n:5d}  {vals}")
        print()


if __name__ == "__main__":
    main()
