"""HDF5 checkpoint I/O with the reference's on-disk layout, over the HDF5 C library through ctypes
(h5py is not in the image; libhdf5 is — the same library the reference links, Makefile:75).

Mirrors src/util.cc:128-208: ``WriteHDF5CPU`` / ``ReadHDF5CPU`` / ``ReadHDF5Shape`` (2-D float32 datasets; a column-major
(rows, cols) matrix is stored as a row-major (cols, rows) dataset, matrix.cc:419-423) and ``WriteHDF5IntAttr`` /
``ReadHDF5IntAttr`` (scalar int attributes on the file root).  A file written here opens in the reference and vice
versa: names are ``<source>:<dest>:weight``, ``…:bias``, ``…:weight_gradient_history`` with attribute ``…:weight_step``
(edge_with_weight.cc:27-64, optimizer.cc:138-156) plus ``__current_iter__`` / ``__lr_reduce_counter__``
(convnet.cc:672-673,744-749)."""
import ctypes
import ctypes.util
import os

import numpy as np

_LIB = None
_T = {}


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    cands = [os.environ.get("CONVNET_HDF5_LIB"), "/opt/conda/lib/libhdf5.so", ctypes.util.find_library("hdf5"), "libhdf5.so"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            _LIB = ctypes.CDLL(c)
            break
        except OSError as e:
            err = e
    if _LIB is None:
        raise ImportError(f"libhdf5 not found (set CONVNET_HDF5_LIB): {err}")
    L = _LIB
    hid, herr, sz = ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
    hs = ctypes.POINTER(ctypes.c_uint64)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype, f.argtypes = res, list(args)

    sig("H5open", herr)
    sig("H5Fcreate", hid, ctypes.c_char_p, ctypes.c_uint, hid, hid)
    sig("H5Fopen", hid, ctypes.c_char_p, ctypes.c_uint, hid)
    sig("H5Fclose", herr, hid)
    sig("H5Screate_simple", hid, ctypes.c_int, hs, hs)
    sig("H5Screate", hid, ctypes.c_int)
    sig("H5Sclose", herr, hid)
    sig("H5Sget_simple_extent_ndims", ctypes.c_int, hid)
    sig("H5Sget_simple_extent_dims", ctypes.c_int, hid, hs, hs)
    sig("H5Dcreate2", hid, hid, ctypes.c_char_p, hid, hid, hid, hid, hid)
    sig("H5Dopen2", hid, hid, ctypes.c_char_p, hid)
    sig("H5Dget_space", hid, hid)
    sig("H5Dwrite", herr, hid, hid, hid, hid, hid, ctypes.c_void_p)
    sig("H5Dread", herr, hid, hid, hid, hid, hid, ctypes.c_void_p)
    sig("H5Dclose", herr, hid)
    sig("H5Lexists", ctypes.c_int, hid, ctypes.c_char_p, hid)
    sig("H5Acreate2", hid, hid, ctypes.c_char_p, hid, hid, hid, hid)
    sig("H5Aopen", hid, hid, ctypes.c_char_p, hid)
    sig("H5Aexists", ctypes.c_int, hid, ctypes.c_char_p)
    sig("H5Awrite", herr, hid, hid, ctypes.c_void_p)
    sig("H5Aread", herr, hid, hid, ctypes.c_void_p)
    sig("H5Aclose", herr, hid)
    sig("H5Eset_auto2", herr, hid, ctypes.c_void_p, ctypes.c_void_p)
    assert L.H5open() >= 0
    L.H5Eset_auto2(0, None, None)      # errors are reported through return codes below, not the library's stderr stack
    _T["float"] = ctypes.c_int64.in_dll(L, "H5T_NATIVE_FLOAT_g").value
    _T["int"] = ctypes.c_int64.in_dll(L, "H5T_NATIVE_INT_g").value
    return L


H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 2, 0, 0, 0


class File:
    """``with File(path, "w") as f`` — H5Fcreate(H5F_ACC_TRUNC) / H5Fopen(H5F_ACC_RDONLY) (convnet.cc:668,741)."""

    def __init__(self, path, mode="r"):
        L = _lib()
        self.path = path
        self.id = (L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT) if mode == "w"
                   else L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT))
        if self.id < 0:
            raise OSError(f"cannot {'create' if mode == 'w' else 'open'} HDF5 file {path}")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.id >= 0:
            _lib().H5Fclose(self.id)
            self.id = -1

    # ---- util.cc:128-138 ----
    def WriteHDF5CPU(self, mat, rows, cols, name):
        """``mat``: the floats in memory order; stored as a (rows, cols) row-major dataset."""
        L = _lib()
        a = np.ascontiguousarray(mat, np.float32).reshape(-1)
        if a.size != rows * cols:
            raise ValueError("Size mismatch")
        dims = (ctypes.c_uint64 * 2)(rows, cols)
        space = L.H5Screate_simple(2, dims, None)
        ds = L.H5Dcreate2(self.id, name.encode(), _T["float"], space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if ds < 0:
            L.H5Sclose(space)
            raise OSError(f"cannot create dataset {name} in {self.path}")
        rc = L.H5Dwrite(ds, _T["float"], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p))
        L.H5Sclose(space)
        L.H5Dclose(ds)
        if rc < 0:
            raise OSError(f"H5Dwrite failed for {name}")

    def Has(self, name):
        return _lib().H5Lexists(self.id, name.encode(), H5P_DEFAULT) > 0

    # ---- util.cc:163-175: (rows, cols) of the column-major matrix = (dims[1] or 1, dims[0]) ----
    def ReadHDF5Shape(self, name):
        L = _lib()
        ds = L.H5Dopen2(self.id, name.encode(), H5P_DEFAULT)
        if ds < 0:
            raise KeyError(f"no dataset {name} in {self.path}")
        space = L.H5Dget_space(ds)
        nd = L.H5Sget_simple_extent_ndims(space)
        dims = (ctypes.c_uint64 * 2)(1, 1)
        L.H5Sget_simple_extent_dims(space, dims, None)
        L.H5Sclose(space)
        L.H5Dclose(ds)
        cols = int(dims[0])
        rows = 1 if nd == 1 else int(dims[1])
        return rows, cols

    # ---- util.cc:188-206 ----
    def ReadHDF5CPU(self, size, name):
        L = _lib()
        rows, cols = self.ReadHDF5Shape(name)
        if rows * cols != size:
            raise ValueError(f"Dimension mismatch: Expected {size} Got {rows}-{cols}")
        out = np.empty(size, np.float32)
        ds = L.H5Dopen2(self.id, name.encode(), H5P_DEFAULT)
        rc = L.H5Dread(ds, _T["float"], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p))
        L.H5Dclose(ds)
        if rc < 0:
            raise OSError(f"H5Dread failed for {name}")
        return out

    # ---- util.cc:177-186 / :188-196 ----
    def WriteHDF5IntAttr(self, name, val):
        L = _lib()
        aid = L.H5Screate(H5S_SCALAR)
        attr = L.H5Acreate2(self.id, name.encode(), _T["int"], aid, H5P_DEFAULT, H5P_DEFAULT)
        v = ctypes.c_int(int(val))
        rc = L.H5Awrite(attr, _T["int"], ctypes.byref(v)) if attr >= 0 else -1
        L.H5Sclose(aid)
        if attr >= 0:
            L.H5Aclose(attr)
        if rc < 0:
            raise OSError(f"cannot write attribute {name}")

    def ReadHDF5IntAttr(self, name, default):
        """Missing attribute: the reference prints a note and leaves the caller's value untouched."""
        L = _lib()
        if L.H5Aexists(self.id, name.encode()) <= 0:
            return default
        attr = L.H5Aopen(self.id, name.encode(), H5P_DEFAULT)
        v = ctypes.c_int(0)
        L.H5Aread(attr, _T["int"], ctypes.byref(v))
        L.H5Aclose(attr)
        return v.value
