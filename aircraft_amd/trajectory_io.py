"""Trajectory files in the reference's on-disk format.

The reference stores trajectories as HDF5: one group per solver iteration, `iteration_<k>/`, holding
`state (13, n)`, `control (7, n)` and `times (n,)` float64 datasets — written by
control/base.py:89-114 (`SaveMixin`, gzip-compressed, plus a `timestamp` attribute) and
main/dynamics/dynamics.py:134-139, read back by plotting/plotting.py:72-95 (`TrajectoryData.load`).
`SaveMixin` names the third dataset `time` while the loader asks for `times`; both names are read
here, `times` is what gets written (so the reference's plotter finds it).

h5py is not part of this image, so this module talks to the HDF5 C library itself through ctypes
(`libhdf5.so`, 1.10 API).  Files written here are ordinary HDF5 — h5py opens them.  Host-side I/O only:
nothing in here touches the GPU.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from dataclasses import dataclass
from datetime import datetime
from typing import List, Optional

import numpy as np

__all__ = ["TrajectoryData", "save_trajectory", "load_trajectory", "list_iterations", "hdf5_available"]

_H5F_ACC_RDONLY, _H5F_ACC_RDWR, _H5F_ACC_TRUNC = 0, 1, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_H5T_FLOAT, _H5T_INTEGER = 1, 0
_hid = C.c_int64
_lib = None


class Hdf5Error(RuntimeError):
    pass


def _candidates():
    env = os.environ.get("AIRCRAFT_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            if "_hl" not in p and "_cpp" not in p and "_fortran" not in p:
                yield p


def _h5():
    """Load libhdf5 once; raise Hdf5Error if the image has none (set AIRCRAFT_HDF5_LIB to point at one)."""
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for path in _candidates():
        try:
            L = C.CDLL(path)
            L.H5open()
            break
        except OSError as e:  # not loadable: try the next candidate
            last = e
    else:
        raise Hdf5Error(f"no HDF5 C library found (set AIRCRAFT_HDF5_LIB); last error: {last}")
    proto = {
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]), "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fclose": (C.c_int, [_hid]), "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]),
        "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Gclose": (C.c_int, [_hid]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]), "H5Ldelete": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Screate": (_hid, [C.c_int]), "H5Sclose": (C.c_int, [_hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Pcreate": (_hid, [_hid]), "H5Pclose": (C.c_int, [_hid]),
        "H5Pset_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(C.c_uint64)]), "H5Pset_deflate": (C.c_int, [_hid, C.c_uint]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]), "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]), "H5Dclose": (C.c_int, [_hid]),
        "H5Tget_class": (C.c_int, [_hid]), "H5Tcopy": (_hid, [_hid]), "H5Tset_size": (C.c_int, [_hid, C.c_size_t]),
        "H5Tclose": (C.c_int, [_hid]),
        "H5Acreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]), "H5Awrite": (C.c_int, [_hid, _hid, C.c_void_p]),
        "H5Aclose": (C.c_int, [_hid]), "H5Aexists": (C.c_int, [_hid, C.c_char_p]), "H5Adelete": (C.c_int, [_hid, C.c_char_p]),
        "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]), "H5Zfilter_avail": (C.c_int, [C.c_int]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, _hid]),
        "H5Gget_num_objs": (C.c_int, [_hid, C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in proto.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.H5Eset_auto2(0, None, None)  # errors come back as status codes; keep the C error stack off stderr
    L.NATIVE_DOUBLE = _hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
    L.IEEE_F64LE = _hid.in_dll(L, "H5T_IEEE_F64LE_g").value
    L.C_S1 = _hid.in_dll(L, "H5T_C_S1_g").value
    L.DATASET_CREATE = _hid.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value
    _lib = L
    return L


def hdf5_available() -> bool:
    try:
        _h5()
        return True
    except Hdf5Error:
        return False


def _ok(v, what):
    if v < 0:
        raise Hdf5Error(f"HDF5 call failed: {what}")
    return v


def _dims(a):
    return (C.c_uint64 * a.ndim)(*a.shape)


def _write_dataset(L, grp, name, data, compress):
    a = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
    bname = name.encode()
    if L.H5Lexists(grp, bname, _H5P_DEFAULT) > 0:  # SaveMixin deletes and rewrites (control/base.py:101-103)
        _ok(L.H5Ldelete(grp, bname, _H5P_DEFAULT), f"delete {name}")
    plist = _H5P_DEFAULT
    if a.ndim == 0:
        space = _ok(L.H5Screate(0), "scalar dataspace")
    else:
        space = _ok(L.H5Screate_simple(a.ndim, _dims(a), None), "dataspace")
        if compress and a.size > 1 and L.H5Zfilter_avail(1) > 0:  # gzip when size > 1 (control/base.py:108-111)
            plist = _ok(L.H5Pcreate(L.DATASET_CREATE), "dcpl")
            _ok(L.H5Pset_chunk(plist, a.ndim, _dims(a)), "chunk")
            _ok(L.H5Pset_deflate(plist, 4), "deflate")
    try:
        ds = _ok(L.H5Dcreate2(grp, bname, L.IEEE_F64LE, space, _H5P_DEFAULT, plist, _H5P_DEFAULT), f"create {name}")
        try:
            _ok(L.H5Dwrite(ds, L.NATIVE_DOUBLE, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)),
                f"write {name}")
        finally:
            L.H5Dclose(ds)
    finally:
        L.H5Sclose(space)
        if plist != _H5P_DEFAULT:
            L.H5Pclose(plist)


def _read_dataset(L, grp, name) -> Optional[np.ndarray]:
    bname = name.encode()
    if L.H5Lexists(grp, bname, _H5P_DEFAULT) <= 0:
        return None
    ds = _ok(L.H5Dopen2(grp, bname, _H5P_DEFAULT), f"open {name}")
    try:
        space = _ok(L.H5Dget_space(ds), "space")
        ftype = _ok(L.H5Dget_type(ds), "type")
        try:
            if L.H5Tget_class(ftype) not in (_H5T_FLOAT, _H5T_INTEGER):
                raise Hdf5Error(f"dataset {name}: not numeric")
            nd = _ok(L.H5Sget_simple_extent_ndims(space), "ndims")
            dims = (C.c_uint64 * max(nd, 1))()
            if nd:
                _ok(L.H5Sget_simple_extent_dims(space, dims, None), "dims")
            out = np.empty(tuple(int(d) for d in dims[:nd]), dtype=np.float64)
            if out.size:  # the library converts whatever numeric type is stored to native double
                _ok(L.H5Dread(ds, L.NATIVE_DOUBLE, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)),
                    f"read {name}")
            return out
        finally:
            L.H5Tclose(ftype)
            L.H5Sclose(space)
    finally:
        L.H5Dclose(ds)


def _write_str_attr(L, obj, name, text):
    raw = text.encode()
    bname = name.encode()
    if L.H5Aexists(obj, bname) > 0:
        L.H5Adelete(obj, bname)
    t = _ok(L.H5Tcopy(L.C_S1), "string type")
    _ok(L.H5Tset_size(t, max(len(raw), 1)), "string size")
    sp = _ok(L.H5Screate(0), "scalar space")
    at = _ok(L.H5Acreate2(obj, bname, t, sp, _H5P_DEFAULT, _H5P_DEFAULT), "attribute")
    buf = C.create_string_buffer(raw, max(len(raw), 1))
    try:
        _ok(L.H5Awrite(at, t, buf), "attribute write")
    finally:
        L.H5Aclose(at)
        L.H5Sclose(sp)
        L.H5Tclose(t)


def _open(L, filepath, mode):
    b = os.fsencode(filepath)
    if mode == "r":
        if not os.path.exists(filepath):
            raise FileNotFoundError(filepath)
        return _ok(L.H5Fopen(b, _H5F_ACC_RDONLY, _H5P_DEFAULT), f"open {filepath}")
    if mode == "w" or not os.path.exists(filepath):
        return _ok(L.H5Fcreate(b, _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT), f"create {filepath}")
    if mode == "a":
        return _ok(L.H5Fopen(b, _H5F_ACC_RDWR, _H5P_DEFAULT), f"open {filepath}")
    raise ValueError(f"mode must be 'r', 'w' or 'a', got {mode!r}")


def save_trajectory(filepath: str, iteration: int, state, control, times=None, *, mode: str = "a",
                    compress: bool = True, extra: Optional[dict] = None) -> None:
    """Write `iteration_<k>/{state, control, times}` (control/base.py:89-105, main/dynamics/dynamics.py:134-139).

    state (13, n), control (7, n) — or any (rows, n) arrays; torch tensors are copied to the host.  `extra` adds
    further named datasets to the group (the reference's loader also looks for `lam`, `mu`, `nu`)."""
    L = _h5()

    def host(a):
        return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)

    f = _open(L, filepath, mode)
    try:
        gname = f"iteration_{int(iteration)}".encode()
        if L.H5Lexists(f, gname, _H5P_DEFAULT) > 0:
            g = _ok(L.H5Gopen2(f, gname, _H5P_DEFAULT), "open group")  # require_group (control/base.py:96)
        else:
            g = _ok(L.H5Gcreate2(f, gname, _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT), "create group")
        try:
            _write_str_attr(L, g, "timestamp", datetime.now().strftime("%Y%m%d_%H%M%S"))
            _write_dataset(L, g, "state", host(state), compress)
            _write_dataset(L, g, "control", host(control), compress)
            if times is not None:
                _write_dataset(L, g, "times", host(times), compress)
            for k, v in (extra or {}).items():
                _write_dataset(L, g, k, host(v), compress)
        finally:
            L.H5Gclose(g)
    finally:
        L.H5Fclose(f)


def list_iterations(filepath: str) -> List[int]:
    """Iteration numbers present in a trajectory file, ascending."""
    L = _h5()
    f = _open(L, filepath, "r")
    try:
        n = C.c_uint64()
        _ok(L.H5Gget_num_objs(f, C.byref(n)), "count")
        out = []
        buf = C.create_string_buffer(256)
        for i in range(n.value):
            ln = L.H5Lget_name_by_idx(f, b".", 0, 0, i, buf, 256, _H5P_DEFAULT)  # H5_INDEX_NAME, H5_ITER_INC
            name = buf.value.decode() if ln > 0 else ""
            if name.startswith("iteration_") and name[10:].lstrip("-").isdigit():
                out.append(int(name[10:]))
        return sorted(out)
    finally:
        L.H5Fclose(f)


@dataclass
class TrajectoryData:
    """reference plotting/plotting.py:61-95: fields are None when the group or dataset is missing."""
    state: Optional[np.ndarray] = None
    control: Optional[np.ndarray] = None
    times: Optional[np.ndarray] = None
    lam: Optional[np.ndarray] = None
    mu: Optional[np.ndarray] = None
    nu: Optional[np.ndarray] = None
    iteration: Optional[int] = None

    def load(self, filepath: str, iteration: int) -> "TrajectoryData":
        L = _h5()
        for k in ("state", "control", "times", "lam", "mu", "nu", "iteration"):
            setattr(self, k, None)
        f = _open(L, filepath, "r")
        try:
            gname = f"iteration_{int(iteration)}".encode()
            if L.H5Lexists(f, gname, _H5P_DEFAULT) <= 0:
                return self  # missing iteration -> everything None (plotting.py:86-94)
            g = _ok(L.H5Gopen2(f, gname, _H5P_DEFAULT), "open group")
            try:
                for k in ("state", "control", "times", "lam", "mu", "nu"):
                    setattr(self, k, _read_dataset(L, g, k))
                if self.times is None:  # SaveMixin's name for it (control/base.py:99)
                    self.times = _read_dataset(L, g, "time")
                self.iteration = int(iteration)
            finally:
                L.H5Gclose(g)
        finally:
            L.H5Fclose(f)
        return self


def load_trajectory(filepath: str, iteration: int) -> TrajectoryData:
    return TrajectoryData().load(filepath, iteration)
