"""Shared helpers: build the SAME case / weights / inputs for the CPU oracle and the HIP engine."""
from __future__ import annotations

import numpy as np

from oracle import lb_oracle as O


def oracle_case(ds, isl=None, dtype=np.float64):
    isl = isl or ds.input_seq_length
    return O.case_builder(
        ds.box, ds.metadata, isl,
        cfg_neighbors={"multiplier": ds.multiplier},
        cfg_model={"isotropic_norm": ds.isotropic_norm, "magnitude_features": getattr(ds, "magnitude_features", False)},
        noise_std=ds.noise_std, external_force_fn=ds.external_force_fn, dtype=dtype)


def hip_case(ds, isl=None, dtype="float64"):
    from lagrangebench_amd.case_setup import case_builder
    isl = isl or ds.input_seq_length
    return case_builder(
        ds.box, ds.metadata, isl,
        cfg_neighbors={"multiplier": ds.multiplier},
        cfg_model={"isotropic_norm": ds.isotropic_norm, "magnitude_features": getattr(ds, "magnitude_features", False)},
        noise_std=ds.noise_std, external_force_fn=ds.force, dtype=dtype)


def feature_widths(ds, isl=None):
    isl = isl or ds.input_seq_length
    dim = len(ds.box)
    K = isl - 1
    node_in = K * dim
    if getattr(ds, "magnitude_features", False):
        node_in += K
    if not any(ds.metadata["periodic_boundary_conditions"]):
        node_in += 2 * dim
    if ds.external_force_fn is not None:
        node_in += dim
    return node_in, dim + 1


def make_params(ds, num_mp_steps=10, seed=1234, decoder_scale=0.01, random_affine=True, latent_size=128, blocks_per_step=2):
    """Haiku-default weights from default_rng(seed); decoder output layer scaled so a random
    net does not blow the neighbor count up.  random_affine additionally randomises biases and
    LayerNorm scale/offset so those code paths are exercised (defaults are 0 / 1 / 0)."""
    node_in, edge_in = feature_widths(ds)
    dim = len(ds.box)
    rng = np.random.default_rng(seed)
    p = O.gns_init(rng, node_in=node_in, edge_in=edge_in, particle_dimension=dim, latent_size=latent_size,
                   blocks_per_step=blocks_per_step, num_mp_steps=num_mp_steps, decoder_scale=decoder_scale)
    if random_affine:
        r2 = np.random.default_rng(seed + 1)
        for k, v in p.items():
            if "b" in v:
                v["b"] = (0.1 * r2.standard_normal(v["b"].shape)).astype(np.float32)
                if k == f"decoder/linear_{blocks_per_step - 1}":
                    v["b"] *= np.float32(decoder_scale)
            if "scale" in v:
                v["scale"] = (1.0 + 0.2 * r2.standard_normal(v["scale"].shape)).astype(np.float32)
                v["offset"] = (0.1 * r2.standard_normal(v["offset"].shape)).astype(np.float32)
    return p


def oracle_model_apply(num_mp_steps):
    def apply(params, state, sample):
        feats, ptype = sample
        return O.gns_apply(params, feats, ptype, num_mp_steps=num_mp_steps, skip_padding=True), state
    return apply


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def elementwise_stats(a, b, floor_frac=1e-3):
    """Element-wise relative error |a-b| / |b| over the entries that are not tiny (|b| > floor_frac * max|b|):
    (99.9th percentile, maximum, number of entries looked at).  The max-norm `rel_err` lets an entry 100x below
    the maximum be off by 1e-3 relative; `north_star` asks for 1e-5 relative PER acceleration."""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    keep = np.abs(b) > floor_frac * np.max(np.abs(b))
    r = np.abs(a[keep] - b[keep]) / np.abs(b[keep])
    if r.size == 0:
        return 0.0, 0.0, 0
    return float(np.percentile(r, 99.9)), float(r.max()), int(r.size)


def make_trained_like_params(ds, num_mp_steps=10, seed=77, decoder_scale=0.01):
    """Weights with the statistics of a TRAINED checkpoint rather than of an initialiser: heavy-tailed matrices
    (Student-t, 3 degrees of freedom, rescaled to the haiku variance), LayerNorm scales spread log-uniformly over
    three decades (1e-2 .. 10) with signs, biases much larger than the weights' scale, offsets of order one."""
    p = make_params(ds, num_mp_steps=num_mp_steps, seed=seed, decoder_scale=decoder_scale, random_affine=False)
    r = np.random.default_rng(seed + 5)
    for k, v in p.items():
        if "w" in v and v["w"].ndim == 2 and not k.startswith("decoder/linear_1"):
            w = r.standard_t(3, size=v["w"].shape)
            w *= (1.0 / np.sqrt(v["w"].shape[0])) / np.sqrt(3.0)      # var of t(3) = 3
            v["w"] = np.clip(w, -4.0, 4.0).astype(np.float32)
        if "b" in v and k != "decoder/linear_1":
            v["b"] = (1.5 * r.standard_normal(v["b"].shape)).astype(np.float32)
        if "scale" in v:
            mag = 10.0 ** r.uniform(-2.0, 1.0, size=v["scale"].shape)
            v["scale"] = (mag * r.choice([-1.0, 1.0], size=mag.shape)).astype(np.float32)
            v["offset"] = r.standard_normal(v["offset"].shape).astype(np.float32)
    return p
