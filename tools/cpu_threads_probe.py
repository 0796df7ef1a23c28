"""Probe: torch-CPU thread count vs time of one reference-shaped GNS forward (bench.py cpu_baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import lb_oracle as O, lb_oracle_torch as OT
from lagrangebench_amd.data import make_case
from tests._common import oracle_case, make_params
ds = make_case("tgv3d", n_trajs=1, extra_seq_length=2)
case = oracle_case(ds); L = 10
params = make_params(ds, num_mp_steps=L)
pt_ = OT.params_to_torch(params)
pos, pt = ds[0]
t0 = time.time(); feats, nbrs = case.allocate_eval((pos[:, :6].astype(np.float64), pt)); t1 = time.time()
feats, nbrs = case.preprocess_eval((pos[:, :6].astype(np.float64), pt), nbrs); t2 = time.time()
print("cores", os.cpu_count(), "allocate %.2fs preprocess_eval %.2fs" % (t1 - t0, t2 - t1), flush=True)
for nt in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    torch.set_num_threads(nt)
    OT.gns_apply(pt_, feats, pt, num_mp_steps=L)
    t = time.time(); OT.gns_apply(pt_, feats, pt, num_mp_steps=L); dt = time.time() - t
    print("threads", nt, "gns forward %.2fs" % dt, flush=True)
