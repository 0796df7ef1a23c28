#!/usr/bin/env python
"""Which dispatches overlapped in time?  python tools/rocpd_overlap.py <results.db> <kernel substring>
For every launch of the named kernel (rocprofv3 --kernel-trace database): the kernels whose [start, end] intersect its own."""
import sqlite3
import sys
from collections import Counter

c = sqlite3.connect(sys.argv[1])
t = {r[0].split("_0000")[0]: r[0] for r in c.execute("select name from sqlite_master where type='table'")}
names = {r[0]: r[1] for r in c.execute(f"select id, display_name from {t['rocpd_info_kernel_symbol']}")}
disp = sorted((s, e, names.get(k, str(k))) for k, s, e in c.execute(f"select kernel_id, start, end from {t['rocpd_kernel_dispatch']}"))
hits, n, before = Counter(), 0, Counter()
for i, (s, e, k) in enumerate(disp):
    if sys.argv[2] not in k or e - s < 5000:
        continue
    n += 1
    for s2, e2, k2 in disp[max(0, i - 12):i + 12]:
        if k2 is not k and s2 < e and e2 > s and (s2, e2, k2) != (s, e, k):
            hits[k2[:60]] += 1
    if i:
        before[disp[i - 1][2][:60]] += 1
print(f"{n} launches of '{sys.argv[2]}'; overlapping kernels:")
for k, v in hits.most_common(8):
    print(f"  {v:5d}  {k}")
print("started right after:")
for k, v in before.most_common(4):
    print(f"  {v:5d}  {k}")
