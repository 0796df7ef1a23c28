#!/usr/bin/env python
"""GPU busy time versus wall span of a rocprofv3 kernel trace (rocpd SQLite): how much of a launch-bound step is idle gaps
between dispatches.   python tools/rocpd_gaps.py <results.db> [skip_first_fraction=0.3]"""
import sqlite3
import sys

db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
c = sqlite3.connect(db)
tab = [r[0] for r in c.execute("select name from sqlite_master where type='table'") if r[0].startswith("rocpd_kernel_dispatch")][0]
d = sorted(c.execute(f"select start, end from {tab}"))
d = d[int(len(d) * skip):]
busy = sum(e - s for s, e in d)
span = d[-1][1] - d[0][0]
gaps = [d[i + 1][0] - d[i][1] for i in range(len(d) - 1)]
gaps_pos = [g for g in gaps if g > 0]
print(f"{len(d)} dispatches: span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({busy / span:.2f}), gaps {sum(gaps_pos) / 1e6:.3f} ms; "
      f"mean kernel {busy / len(d) / 1e3:.2f} us, mean gap {sum(gaps_pos) / max(len(gaps_pos), 1) / 1e3:.2f} us, "
      f"gaps > 20 us: {sum(1 for g in gaps if g > 20000)} ({sum(g for g in gaps if g > 20000) / 1e6:.3f} ms)")
