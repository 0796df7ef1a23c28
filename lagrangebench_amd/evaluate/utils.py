"""Rollout export helpers - mirror of lagrangebench/evaluate/utils.py:1-77.

``write_vtk`` stores one particle frame as a legacy-VTK POLYDATA file ParaView opens.  The
reference goes through pyvista (``pyvista.PolyData(r)``, one point-data array per extra key,
``.save(path)``); pyvista is used here too when it is importable, otherwise an equivalent
ASCII legacy file (POINTS + one VERTICES cell per point + POINT_DATA arrays) is written directly.
Host-side IO only - nothing here touches the device.
"""
from __future__ import annotations

import os
import pickle

import numpy as np


def _write_legacy_ascii(r3: np.ndarray, arrays: dict, path: str) -> None:
    n = r3.shape[0]
    with open(path, "w") as f:
        f.write("# vtk DataFile Version 3.0\nlagrangebench particles\nASCII\nDATASET POLYDATA\n")
        f.write(f"POINTS {n} double\n")
        np.savetxt(f, r3, fmt="%.17g")
        f.write(f"VERTICES {n} {2 * n}\n")
        np.savetxt(f, np.stack([np.ones(n, np.int64), np.arange(n, dtype=np.int64)], axis=1), fmt="%d")
        if arrays:
            f.write(f"POINT_DATA {n}\n")
        for k, v in arrays.items():
            v = np.asarray(v)
            is_int = np.issubdtype(v.dtype, np.integer)
            kind, fmt = ("int", "%d") if is_int else ("double", "%.17g")
            if v.ndim == 1:
                f.write(f"SCALARS {k} {kind} 1\nLOOKUP_TABLE default\n")
                np.savetxt(f, v.reshape(-1, 1), fmt=fmt)
            elif v.ndim == 2 and v.shape[1] == 3:
                f.write(f"VECTORS {k} {kind}\n")
                np.savetxt(f, v, fmt=fmt)
            else:
                v2 = v.reshape(n, -1)
                f.write(f"FIELD {k}_field 1\n{k} {v2.shape[1]} {n} {kind}\n")
                np.savetxt(f, v2, fmt=fmt)


def write_vtk(data_dict, path):
    """Store a .vtk file for ParaView - evaluate/utils.py:9-38.  ``data_dict["r"]`` (N, dim)
    positions; every other key becomes a point-data array (2D vectors are zero-padded to 3D)."""
    r = np.asarray(data_dict["r"])
    N, dim = r.shape
    if dim == 2:
        r = np.hstack([r, np.zeros((N, 1))])
    arrays = {}
    for k, v in data_dict.items():
        if k == "r":
            continue
        v = np.asarray(v)
        if dim == 2 and v.ndim == 2:
            v = np.hstack([v, np.zeros((N, 1))])
        arrays[k] = v
    try:
        import pyvista
    except ImportError:
        _write_legacy_ascii(np.asarray(r, np.float64), arrays, path)
        return
    data_pv = pyvista.PolyData(r)
    for k, v in arrays.items():
        data_pv[k] = v
    data_pv.save(path)


def pkl2vtk(src_path, dst_path=None):
    """Convert a rollout pickle file to a set of vtk files - evaluate/utils.py:41-77:
    ``<prefix>_<k>.vtk`` for the predicted frames and ``<prefix>_ref_<k>.vtk`` for the ground truth."""
    if dst_path is None:
        dst_path = os.path.dirname(src_path)
    os.makedirs(dst_path, exist_ok=True)
    with open(src_path, "rb") as f:
        rollout = pickle.load(f)
    file_prefix = os.path.join(dst_path, os.path.basename(src_path).split(".")[0])
    for k in range(rollout["predicted_rollout"].shape[0]):
        write_vtk({"r": rollout["predicted_rollout"][k], "tag": rollout["particle_type"]}, f"{file_prefix}_{k}.vtk")
        write_vtk({"r": rollout["ground_truth_rollout"][k], "tag": rollout["particle_type"]},
                  f"{file_prefix}_ref_{k}.vtk")
