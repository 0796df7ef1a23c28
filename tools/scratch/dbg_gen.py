import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import segnn_irreps_oracle as G, segnn_oracle as S
from tests._common import hip_case, oracle_case, rel_err
from tests.test_segnn_irreps import _gpu_setup, _np
for cfg in [("small2d", 1.0, 2, 1, 2, "batch", 2, 64), ("small3d", 1.0, 2, 1, 1, "batch", 2, 64), ("ldc3d", 0.5, 2, 2, 2, "instance", 2, 64),
            ("small3d", 1.0, 2, 1, 1, None, 3, 32), ("small2d", 1.0, 2, 0, 1, None, 1, 64)]:
    name, scale, L, lh, la, norm, blocks, units = cfg
    ds, model, params, homog = _gpu_setup(*cfg)
    ocase, hcase = oracle_case(ds), hip_case(ds)
    isl = ds.input_seq_length
    pos = np.stack([ds[0][0], ds[1][0]]); pt = np.stack([ds[0][1], ds[1][1]])
    feats, _ = hcase.allocate_eval((pos[:, :, :isl], pt))
    handle = model.handle(feats.engine, params)
    tap = handle.set_tap(True)
    pred, _ = model.apply(params, {}, (feats, pt))
    acc, tap = _np(pred["acc"]), _np(tap)
    N = pos.shape[1]; hdim = G.dim_of(params["hidden"])
    for b in range(2):
        of, _ = ocase.allocate_eval((pos[b][:, :isl].astype(np.float64), pt[b]))
        ref, lat = G.segnn_apply(params, dict(of), pt[b], isl - 1, homog, return_latents=True)
        with S.precision(np.float64):
            ref64, lat64 = G.segnn_apply(params, dict(of), pt[b], isl - 1, homog, return_latents=True)
        for k in range(len(lat)):
            got = tap[k][b*N:(b+1)*N][:, :hdim]
            print(cfg, b, k, "dev-vs-f32 %.2e dev-vs-f64 %.2e f32-vs-f64 %.2e" % (rel_err(got, lat[k]), rel_err(got, lat64[k]), rel_err(lat[k], lat64[k])))
        print(cfg, b, "acc", "dev-vs-f32 %.2e dev-vs-f64 %.2e f32-vs-f64 %.2e" % (rel_err(acc[b], ref["acc"]), rel_err(acc[b], ref64["acc"]), rel_err(ref["acc"], ref64["acc"])))
    handle.set_tap(False)
