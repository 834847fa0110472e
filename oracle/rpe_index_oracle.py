"""numpy/ctypes front end of oracle/rpe_index_oracle.c (TEST INFRASTRUCTURE).

fwd/bwd take and return numpy arrays.  16-bit floats are handled as raw uint16 patterns
for the forward (a pure copy) and via float32 accumulation + one rounding in the backward
(the documented behaviour of the HIP kernel for f16/bf16; the reference accumulates in
half precision in a thread-dependent order, rpe_index.cpp:75-80, so there is nothing
bit-exact to follow there).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "librpe_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "rpe_index_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_I64 = ctypes.c_int64


def fwd(inp, idx):
    """Y[b,h,i,j] = inp[b,h,i,idx[i,j]] for a C-contiguous inp (B,H,Lq,nb)."""
    inp = np.ascontiguousarray(inp)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, H, Lq, nb = inp.shape
    Lq2, Lk = idx.shape
    assert Lq2 == Lq
    width = inp.dtype.itemsize
    raw = {2: np.uint16, 4: np.uint32, 8: np.uint64}[width]
    fn = getattr(_load(), {2: "oracle_rpe_index_fwd_u16", 4: "oracle_rpe_index_fwd_u32",
                           8: "oracle_rpe_index_fwd_u64"}[width])
    y = np.empty((B, H, Lq, Lk), dtype=raw)
    fn(_p(y), _p(inp.view(raw)), _p(idx), _I64(B), _I64(H), _I64(Lq), _I64(Lk), _I64(nb))
    return y.view(inp.dtype)


def fwd_strided_f32(storage, shape, strides, idx):
    """Forward on a strided float32 view: element strides (s0..s3) into flat `storage`."""
    storage = np.ascontiguousarray(storage, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, H, Lq, _ = shape
    Lk = idx.shape[1]
    y = np.empty((B, H, Lq, Lk), dtype=np.uint32)
    _load().oracle_rpe_index_fwd_strided_u32(
        _p(y), _p(storage.view(np.uint32)), _p(idx), _I64(B), _I64(H), _I64(Lq), _I64(Lk),
        *[_I64(int(s)) for s in strides])
    return y.view(np.float32)


def bwd(gout, idx, nb, gin=None):
    """gin[b,h,i,u] (+)= sum_j [idx[i,j]==u] gout[b,h,i,j], addends in ascending j.
    float32 / float64 exactly as the sequential reference loop; other dtypes must be
    passed already widened to float32 (see module docstring)."""
    gout = np.ascontiguousarray(gout)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, H, Lq, Lk = gout.shape
    if gin is None:
        gin = np.zeros((B, H, Lq, nb), dtype=gout.dtype)
    else:
        gin = np.array(gin, dtype=gout.dtype, order="C", copy=True)
    fn = {np.dtype(np.float32): "oracle_rpe_index_bwd_f32",
          np.dtype(np.float64): "oracle_rpe_index_bwd_f64"}[gout.dtype]
    getattr(_load(), fn)(_p(gin), _p(gout), _p(idx), _I64(B), _I64(H), _I64(Lq), _I64(Lk), _I64(nb))
    return gin
