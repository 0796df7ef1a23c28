"""Average duration of the k_dw_part* / k_lin32h launches (or of the kernels whose name contains argv[2]) of a rocprofv3 rocpd
database, grouped by (kernel, grid): separates the edge-sized from the node-sized calls of one kernel
(python tools/rocpd_by_grid.py <results.db> [name part])."""
import sqlite3, glob, collections, sys
db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t or "info_kernel_symbol" in t]
sym = ks[0]
cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in cols else cols[-1]
rows = c.execute(f"select s.{name_col}, d.start, d.end, d.grid_size_x, d.group_segment_size from {kd} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
agg = collections.defaultdict(list)
for n, s, e, gx, gy in rows:
    if (sys.argv[2] in n) if len(sys.argv) > 2 else ("k_dw_part" in n or "k_lin32h" in n):
        agg[(n[:28], gx, gy)].append((e - s) / 1000)
for k, v in sorted(agg.items()):
    print(k, len(v), round(sum(v) / len(v), 2))
