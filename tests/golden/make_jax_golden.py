"""Pin the NETWORK layer of the oracle to the real reference: run lagrangebench (JAX / Haiku / jraph /
e3nn-jax / jax-md) once and dump inputs, weights and outputs as small .npz fixtures.

    python tests/golden/make_jax_golden.py [--ref /root/reference] [--out tests/golden]

Needs an environment in which the reference imports (jax, jaxlib, dm-haiku, jraph, jax-sph,
e3nn-jax, omegaconf - the pins of the reference's pyproject.toml).  None of them is installable in
the build container (no network, no wheels), so the fixtures `jax_gns_*.npz` / `jax_segnn_*.npz` do
not exist yet and the network layer of the oracle is "parity unpinned" (DESIGN.md section 2).  The
moment someone runs this script on a machine with JAX and commits the .npz files,
`tests/test_jax_golden.py` picks them up: the CPU oracle is checked against them in the
`-m "not gpu"` suite and the HIP engine in the `-m gpu` suite.  Only DATA is written (positions,
weights, outputs); no reference source is copied.

What is dumped per case (all through the reference's public API, cited by file:line):
  * a seeded particle cloud (N, T, dim) f32 + particle types + the metadata dict (JSON)
  * case_builder(...) -> allocate_eval -> features dict + neighbors.idx (case_setup/case.py:62-269)
  * hk.transform_with_state(GNS / SEGNN) params flattened as "<module>//<leaf>" -> array
    (models/gns.py:35-171, models/segnn.py:403-610, runner.py:192-292)
  * model.apply(params, state, (features, particle_type)) -> "acc"
  * case.integrate(pred, positions) -> next position (case.py:230-259)
  * a 5-step _eval_batched_rollout prediction (evaluate/rollout.py:78-178)
Round 4 - one extra fixture, jax_extras.npz, pins what is still [mem] besides the networks:
  * MetricsComputer(["sinkhorn"], ot_backend="ott" | "pot") on a strided pair of rollouts (evaluate/metrics.py:127-213)
  * a neighbor list with SEVERAL particles per cell (cell capacity >= 3) in jax-md's raw slot order
    (case_setup/case.py:120-130): the reference's only golden list (tests/case_test.py:77-82) has capacity 1 and cannot
    see the slot rotation / stencil order
  * the GNS parameters written by the reference's own save_haiku (utils.py:61-91): the files' bytes, so that the
    checkpoint reader (lagrangebench_amd/utils.py: load_haiku) is pinned to real module names and file layout
  * get_dataset_stats (data/utils.py:9-45), anisotropic and isotropic
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _cloud(n_side, dim, dx, isl, n_extra, seed, pbc=True):
    """Jittered lattice advected by a smooth field: the same construction as
    lagrangebench_amd/data/synthetic.py (small2d / small3d), kept independent of the product code."""
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * dim, indexing="ij"), -1).reshape(-1, dim)
    box = np.full(dim, n_side * dx)
    p = np.mod((g + 0.5) * dx + rng.normal(0, 0.1 * dx, g.shape), box)
    T = isl + n_extra
    out = np.empty((len(p), T, dim), np.float32)
    k = 2 * np.pi / box[0]
    for t in range(T):
        out[:, t] = p
        u = np.zeros_like(p)
        u[:, 0] = np.sin(k * p[:, 0]) * np.cos(k * p[:, 1])
        u[:, 1] = -np.cos(k * p[:, 0]) * np.sin(k * p[:, 1])
        p = np.mod(p + 0.3 * dx * u + rng.normal(0, 0.01 * dx, p.shape), box)
    return out, box


def _metadata(dim, n, box, dx, rc, T):
    return {"case": "golden", "dim": dim, "dx": dx, "dt": 1.0, "write_every": 1, "num_particles_max": n,
            "periodic_boundary_conditions": [True] * dim, "bounds": [[0.0, float(b)] for b in box],
            "default_connectivity_radius": rc, "sequence_length_train": T, "sequence_length_test": T,
            "vel_mean": [0.0] * dim, "vel_std": [0.3 * dx] * dim, "acc_mean": [0.0] * dim,
            "acc_std": [0.04 * dx] * dim}


def _flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        key = f"{prefix}//{k}" if prefix else k
        if isinstance(v, dict) or hasattr(v, "items"):
            out.update(_flatten(v, key))
        else:
            out[key] = np.asarray(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    sys.path.insert(0, a.ref)
    try:
        import jax
        jax.config.update("jax_enable_x64", True)  # runner.py:35-36 (cfg.dtype == "float64")
        import haiku as hk
        import jax.numpy as jnp
        import jmp
        import lagrangebench
        from lagrangebench import models
        from lagrangebench.evaluate.rollout import _eval_batched_rollout, _forward_eval
        from lagrangebench.models.utils import node_irreps
        from lagrangebench.utils import NodeType, broadcast_from_batch
    except ImportError as exc:
        print(f"make_jax_golden: the reference does not import here ({exc}); nothing written.")
        return 2

    from functools import partial

    for tag, dim, n_side, L in [("2d", 2, 14, 3), ("3d", 3, 8, 2)]:
        isl, n_extra, dx = 6, 5, 0.05
        rc = 1.45 * dx
        pos, box = _cloud(n_side, dim, dx, isl, n_extra, seed=dim)
        n = pos.shape[0]
        ptype = np.zeros(n, np.int32)
        ptype[: n // 8] = 1  # some SOLID_WALL particles: embedding rows + kinematic mask are exercised
        md = _metadata(dim, n, box, dx, rc, isl + n_extra)
        case = lagrangebench.case_builder(box=box, metadata=md, input_seq_length=isl,
                                          cfg_neighbors={"backend": "jaxmd_vmap", "multiplier": 1.25},
                                          cfg_model={"isotropic_norm": False, "magnitude_features": False},
                                          noise_std=3e-4, external_force_fn=None, dtype=jnp.float64)
        sample = (jnp.asarray(pos[:, :isl]), jnp.asarray(ptype))
        features, nbrs = case.allocate_eval(sample)

        def dump(model_name, model_fn, fname, extra=None):
            model = hk.without_apply_rng(hk.transform_with_state(model_fn))
            policy = jmp.get_policy("params=float32,compute=float32,output=float32")  # runner.py:71-72
            hk.mixed_precision.set_policy(getattr(models, model_name), policy)
            params, state = model.init(jax.random.PRNGKey(7), (features, sample[1]))
            pred, _ = model.apply(params, state, (features, sample[1]))
            nxt = case.integrate(pred, sample[0])
            fwd = jax.vmap(partial(_forward_eval, model_apply=model.apply, case_integrate=case.integrate),
                           in_axes=(None, None, 0, 0, 0))
            pre = jax.vmap(case.preprocess_eval, in_axes=(0, 0))
            roll, _, _ = _eval_batched_rollout(
                forward_eval_vmap=fwd, preprocess_eval_vmap=pre, case=case, params=params, state=state,
                traj_batch_i=(jnp.asarray(pos[None]), jnp.asarray(ptype[None])), neighbors=nbrs,
                metrics_computer_vmap=lambda p, t: {}, n_rollout_steps=n_extra, t_window=isl)
            out = {"position": pos, "particle_type": ptype, "metadata_json": np.array(json.dumps(md)),
                   "idx": np.asarray(nbrs.idx), "acc": np.asarray(pred["acc"]), "next_position": np.asarray(nxt),
                   "rollout": np.asarray(roll)[0], "num_mp_steps": np.array(L)}
            out.update(extra or {})
            for k, v in features.items():
                out[f"feat//{k}"] = np.asarray(v)
            for k, v in _flatten(hk.data_structures.to_mutable_dict(params)).items():
                out[f"param//{k}"] = v
            np.savez_compressed(os.path.join(a.out, fname), **out)
            print("wrote", fname, {k: v.shape for k, v in out.items() if k.startswith("param//")}.__len__(), "leaves")

        dump("GNS", lambda x: models.GNS(particle_dimension=dim, latent_size=128, blocks_per_step=2,
                                         num_mp_steps=L, num_particle_types=NodeType.SIZE,
                                         particle_type_embedding_size=16)(x), f"jax_gns_{tag}.npz")
        try:
            from e3nn_jax import Irreps
            irr = node_irreps(md, isl, False, False, False)
            # the shipped switches, then the ones csrc/lb_segnn_gen.hip runs (round 5): lmax 2 (pins the SIGNS of the real 3j
            # symbols and the l = 2 harmonics: oracle/segnn_irreps_oracle.py A7 - A9) and e3nn's BatchNorm as the reference
            # calls it (A10: training-mode statistics, eps, instance = statistics over an axis of length one)
            for suffix, lh, la, norm in (("", 1, 1, "none"), ("_l22", 2, 2, "none"), ("_l21", 2, 1, "none"),
                                         ("_bn", 1, 1, "batch"), ("_in", 1, 1, "instance"), ("_l22bn", 2, 2, "batch")):
                dump("SEGNN", lambda x, lh=lh, la=la, norm=norm: models.SEGNN(
                    node_features_irreps=irr, edge_features_irreps=Irreps("1x1o + 1x0e"), scalar_units=64, lmax_hidden=lh,
                    lmax_attributes=la, output_irreps=Irreps("1x1o"), num_mp_steps=L, n_vels=isl - 1,
                    velocity_aggregate="avg", homogeneous_particles=False, blocks_per_step=2, norm=norm)(x),
                     f"jax_segnn{suffix}_{tag}.npz", extra={"lmax_hidden": np.array(lh), "lmax_attributes": np.array(la),
                                                           "norm": np.array(norm)})
        except ImportError as exc:
            print("e3nn_jax missing, SEGNN fixture skipped:", exc)
    extras(a, jax, jnp, hk, jmp, lagrangebench, models, NodeType)
    return 0


def extras(a, jax, jnp, hk, jmp, lagrangebench, models, NodeType):
    """jax_extras.npz: sinkhorn (ott / pot), a dense-cell neighbor list in raw slot order, a save_haiku checkpoint's
    bytes, get_dataset_stats."""
    import tempfile
    from lagrangebench.data.utils import get_dataset_stats
    from lagrangebench.evaluate.metrics import MetricsComputer
    from lagrangebench.utils import save_haiku
    out = {}
    # ---- dense cells: dx = 0.02, r_c = 3 dx -> ~9 particles per 2D cell
    dim, n_side, isl, n_extra, dx = 2, 18, 6, 21, 0.02
    rc = 3.0 * dx
    pos, box = _cloud(n_side, dim, dx, isl, n_extra, seed=11)
    n = pos.shape[0]
    ptype = np.zeros(n, np.int32)
    md = _metadata(dim, n, box, dx, rc, isl + n_extra)
    case = lagrangebench.case_builder(box=box, metadata=md, input_seq_length=isl,
                                      cfg_neighbors={"backend": "jaxmd_vmap", "multiplier": 1.25},
                                      cfg_model={"isotropic_norm": False, "magnitude_features": False},
                                      noise_std=3e-4, external_force_fn=None, dtype=jnp.float64)
    features, nbrs = case.allocate_eval((jnp.asarray(pos[:, :isl]), jnp.asarray(ptype)))
    out["dense//position"] = pos
    out["dense//metadata_json"] = np.array(json.dumps(md))
    out["dense//idx"] = np.asarray(nbrs.idx)                      # raw jax-md order, padding = n
    out["dense//cell_capacity"] = np.array(int(getattr(nbrs, "cell_list_capacity", 0) or 0))
    out["dense//rel_disp"] = np.asarray(features["rel_disp"])
    # the update path on a later frame (same capacities): preprocess_eval
    f2, n2 = case.preprocess_eval((jnp.asarray(pos[:, 3:3 + isl]), jnp.asarray(ptype)), nbrs)
    out["dense//idx_update"] = np.asarray(n2.idx)
    # ---- sinkhorn: a 21-frame "prediction" (the cloud) against a perturbed "target", stride 10 -> frames 0, 10, 20
    rng = np.random.default_rng(5)
    pred = np.transpose(pos[:, isl:], (1, 0, 2)).astype(np.float64)          # (T, N, dim)
    targ = np.mod(pred + rng.normal(0, 0.4 * dx, pred.shape), box)
    out["sinkhorn//pred"], out["sinkhorn//target"] = pred, targ
    for backend in ("ott", "pot"):
        try:
            mc = MetricsComputer(["sinkhorn"], case.displacement, md, isl, stride=10, ot_backend=backend)
            out[f"sinkhorn//{backend}"] = np.asarray(mc(jnp.asarray(pred), jnp.asarray(targ))["sinkhorn"])
        except Exception as exc:  # POT is an optional import of the reference
            print(f"sinkhorn backend {backend} skipped: {exc!r}")
    # ---- get_dataset_stats
    md_aniso = dict(md, vel_mean=[0.01, -0.02], vel_std=[0.3 * dx, 0.5 * dx], acc_mean=[1e-4, 2e-4], acc_std=[0.04 * dx, 0.02 * dx])
    for iso in (False, True):
        st = get_dataset_stats(md_aniso, iso, 3e-4)
        for q in ("acceleration", "velocity"):
            for m in ("mean", "std"):
                out[f"stats//iso{int(iso)}//{q}//{m}"] = np.asarray(st[q][m])
    out["stats//metadata_json"] = np.array(json.dumps(md_aniso))
    # ---- a checkpoint written by the reference's save_haiku
    L = 2
    model = hk.without_apply_rng(hk.transform_with_state(
        lambda x: models.GNS(particle_dimension=dim, latent_size=128, blocks_per_step=2, num_mp_steps=L,
                             num_particle_types=NodeType.SIZE, particle_type_embedding_size=16)(x)))
    hk.mixed_precision.set_policy(models.GNS, jmp.get_policy("params=float32,compute=float32,output=float32"))
    params, state = model.init(jax.random.PRNGKey(3), (features, jnp.asarray(ptype)))
    pred_acc, _ = model.apply(params, state, (features, jnp.asarray(ptype)))
    with tempfile.TemporaryDirectory() as d:
        ck = os.path.join(d, "ckp")
        os.makedirs(ck)
        save_haiku(ck, params, state, None, {"step": 7, "loss": 0.5})
        for fn in sorted(os.listdir(ck)):
            fp = os.path.join(ck, fn)
            if os.path.isfile(fp):
                out[f"ckpt//{fn}"] = np.frombuffer(open(fp, "rb").read(), dtype=np.uint8)
    out["ckpt//acc"] = np.asarray(pred_acc["acc"])
    out["ckpt//num_mp_steps"] = np.array(L)
    for k, v in features.items():
        out[f"ckpt//feat//{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(a.out, "jax_extras.npz"), **out)
    print("wrote jax_extras.npz", sorted(out))


if __name__ == "__main__":
    sys.exit(main())
