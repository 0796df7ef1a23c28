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


def hip_case(ds, isl=None):
    from lagrangebench_amd.case_setup import case_builder
    isl = isl or ds.input_seq_length
    return case_builder(
        ds.box, ds.metadata, isl,
        cfg_neighbors={"multiplier": ds.multiplier},
        cfg_model={"isotropic_norm": ds.isotropic_norm, "magnitude_features": getattr(ds, "magnitude_features", False)},
        noise_std=ds.noise_std, external_force_fn=ds.force)


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


def make_params(ds, num_mp_steps=10, seed=1234, decoder_scale=0.01, random_affine=True):
    """Haiku-default weights from default_rng(seed); decoder output layer scaled so a random
    net does not blow the neighbor count up.  random_affine additionally randomises biases and
    LayerNorm scale/offset so those code paths are exercised (defaults are 0 / 1 / 0)."""
    node_in, edge_in = feature_widths(ds)
    dim = len(ds.box)
    rng = np.random.default_rng(seed)
    p = O.gns_init(rng, node_in=node_in, edge_in=edge_in, particle_dimension=dim,
                   num_mp_steps=num_mp_steps, decoder_scale=decoder_scale)
    if random_affine:
        r2 = np.random.default_rng(seed + 1)
        for k, v in p.items():
            if "b" in v:
                v["b"] = (0.1 * r2.standard_normal(v["b"].shape)).astype(np.float32)
                if k == "decoder/linear_1":
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
