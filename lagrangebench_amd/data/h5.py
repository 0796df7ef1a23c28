"""Minimal HDF5 reader for LagrangeBench datasets.

The reference reads its datasets with h5py (lagrangebench/data/data.py:103-148,199-225); h5py is
not available here, so this module binds the HDF5 C library directly with ctypes (h5py is used
instead when it happens to be importable).  Only what H5Dataset needs: list the root groups,
dataset shapes, full and hyperslab reads of float32 / int32 datasets (chunked + deflate handled
by the library).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = []
    env = os.environ.get("LB_HDF5_LIB")
    if env:
        cands.append(env)
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/libhdf5.so*"):
        cands += sorted(glob.glob(pat))
    err = None
    for c in cands:
        try:
            lib = C.CDLL(c)
            lib.H5open.restype = herr_t
            if lib.H5open() < 0:
                continue
            _bind(lib)
            _lib = lib
            return lib
        except OSError as e:  # try the next candidate
            err = e
    raise ImportError("no usable HDF5 library found (set LB_HDF5_LIB=/path/to/libhdf5.so or install h5py)"
                      + (f": {err}" if err else ""))


def _bind(lib):
    sig = {
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Gget_info": (herr_t, [hid_t, C.POINTER(_GInfo)]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t),
                                        C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args


def _native(lib, np_dtype) -> int:
    name = {np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
            np.dtype(np.int32): "H5T_NATIVE_INT_g", np.dtype(np.int64): "H5T_NATIVE_LLONG_g"}[np.dtype(np_dtype)]
    return hid_t.in_dll(lib, name).value


class Dataset:
    def __init__(self, f: "File", path: str):
        self._f, self.path = f, path
        lib = f._lib
        self._id = lib.H5Dopen2(f._id, path.encode(), 0)
        if self._id < 0:
            raise KeyError(path)
        sid = lib.H5Dget_space(self._id)
        nd = lib.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        lib.H5Sget_simple_extent_dims(sid, dims, None)
        lib.H5Sclose(sid)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        tid = lib.H5Dget_type(self._id)
        cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
        lib.H5Tclose(tid)
        # H5T_INTEGER = 0, H5T_FLOAT = 1
        self.dtype = np.dtype({(1, 4): np.float32, (1, 8): np.float64, (0, 4): np.int32, (0, 8): np.int64}[(cls, size)])

    def read(self, start: Optional[Sequence[int]] = None, count: Optional[Sequence[int]] = None) -> np.ndarray:
        lib = self._f._lib
        nd = len(self.shape)
        start = [0] * nd if start is None else list(start)
        count = [self.shape[i] - start[i] for i in range(nd)] if count is None else list(count)
        out = np.empty(count, dtype=self.dtype)
        if out.size == 0:
            return out
        fs = lib.H5Dget_space(self._id)
        st, ct = (hsize_t * nd)(*start), (hsize_t * nd)(*count)
        if lib.H5Sselect_hyperslab(fs, 0, st, None, ct, None) < 0:
            raise IOError(f"hyperslab selection failed on {self.path}")
        ms = lib.H5Screate_simple(nd, ct, None)
        rc = lib.H5Dread(self._id, _native(lib, self.dtype), ms, fs, 0, out.ctypes.data_as(C.c_void_p))
        lib.H5Sclose(ms)
        lib.H5Sclose(fs)
        if rc < 0:
            raise IOError(f"H5Dread failed on {self.path}")
        return out

    def __getitem__(self, key):
        """Full read for ``[:]`` / ``[...]``; a leading slice ``[a:b]`` becomes a hyperslab read."""
        if key is Ellipsis or key == slice(None):
            return self.read()
        if isinstance(key, slice):
            a, b, step = key.indices(self.shape[0])
            if step != 1:
                return self.read()[key]
            nd = len(self.shape)
            return self.read([a] + [0] * (nd - 1), [max(b - a, 0)] + list(self.shape[1:]))
        return self.read()[key]

    def close(self):
        if self._id >= 0:
            self._f._lib.H5Dclose(self._id)
            self._id = -1


class File:
    """``with File(path) as f: f.keys(); f["00000/position"][a:b]`` - the h5py subset H5Dataset uses."""

    def __init__(self, path: str, mode: str = "r"):
        assert mode == "r", "read-only"
        self._lib = _load()
        self._id = self._lib.H5Fopen(path.encode(), 0, 0)
        if self._id < 0:
            raise FileNotFoundError(path)
        self._open: dict = {}  # path -> Dataset: one HDF5 dataset id per path for the file's lifetime

    def keys(self) -> List[str]:
        info = _GInfo()
        if self._lib.H5Gget_info(self._id, C.byref(info)) < 0:
            raise IOError("H5Gget_info failed")
        out = []
        for i in range(int(info.nlinks)):
            n = self._lib.H5Lget_name_by_idx(self._id, b".", 0, 0, i, None, 0, 0)
            buf = C.create_string_buffer(n + 1)
            self._lib.H5Lget_name_by_idx(self._id, b".", 0, 0, i, buf, n + 1, 0)
            out.append(buf.value.decode())
        return out

    def __getitem__(self, path: str) -> Dataset:
        d = self._open.get(path)
        if d is None:
            d = self._open[path] = Dataset(self, path)
        return d

    def close(self):
        for d in self._open.values():
            d.close()
        self._open = {}
        if self._id >= 0:
            self._lib.H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def open_file(path: str):
    """h5py.File if h5py is importable, else the ctypes reader above (same subset of the API)."""
    try:
        import h5py  # type: ignore
        return h5py.File(path, "r")
    except ImportError:
        return File(path, "r")
