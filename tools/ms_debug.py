"""Debug helper: compare the engine's per-layer node latents with the oracle on a small case, print the error
per feature block / row block (layout bugs show up as a pattern)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lagrangebench_amd.data import make_case
from lagrangebench_amd.models import GNS
from oracle import lb_oracle as O
from tests._common import hip_case, make_params, oracle_case

name, L = (sys.argv[1] if len(sys.argv) > 1 else "small2d"), int(sys.argv[2]) if len(sys.argv) > 2 else 3
ds = make_case(name, n_trajs=1, extra_seq_length=3)
ocase, hcase = oracle_case(ds), hip_case(ds)
isl, dim = ds.input_seq_length, len(ds.box)
params = make_params(ds, num_mp_steps=L, decoder_scale=1.0)
model = GNS(dim, 128, 2, L, 16)
pos, pt = ds[0][0][None], ds[0][1][None]
feats, nbrs = hcase.allocate_eval((pos[:, :, :isl], pt))
eng = feats.engine
handle = model.handle(eng, params)
NOTAP = os.environ.get("MS_DEBUG_NOTAP") == "1"   # the persistent processor launch only runs without a tap
tap = None if NOTAP else handle.set_tap(True)
pred, _ = model.apply(params, {}, (feats, pt))
pred2, _ = model.apply(params, {}, (feats, pt))
print("repeatable:", bool((pred["acc"] == pred2["acc"]).all()), eng.kernel_names())
tap = None if NOTAP else tap.cpu().numpy()
of, on = ocase.allocate_eval((pos[0][:, :isl].astype(np.float64), pt[0]))
ref, inter = O.gns_apply(params, of, pt[0], num_mp_steps=L, skip_padding=True, return_intermediates=True)
N = pos.shape[1]
names = ["enc_n"] + [f"n{k}" for k in range(L)]
ONLY_LAST = os.environ.get("LB_PERSIST_TAP") == "1"
for i, nm in enumerate([] if NOTAP else names):
    if ONLY_LAST and i != L:
        continue
    d = np.abs(tap[i][:N] - inter[nm])
    print(nm, "max err", d.max(), "ref max", np.abs(inter[nm]).max())
    print("  per 16-feature block:", np.round(d.reshape(N, 8, 16).max(axis=(0, 2)), 5))
    print("  per row%16:", np.round(d.reshape(-1, 16, 128)[: N // 16].max(axis=(0, 2)), 5))
print("acc err", np.abs(pred["acc"].cpu().numpy()[0] - ref["acc"]).max(), "ref max", np.abs(ref["acc"]).max())
