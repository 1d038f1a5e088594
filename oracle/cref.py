"""ORACLE (test infrastructure) -- ctypes binding of oracle/liborc.so (literal-definition C loops)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def conv2d_same_fwd(x, w, b=None):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    y = np.empty((n, h, wd, co), np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    lib().orc_conv2d_same_fwd_f32(_p(x), _p(w), _p(bb), _p(y), n, h, wd, ci, co, kh, kw)
    return y


def conv2d_same_bwd_input(dy, w, ci):
    dy = np.ascontiguousarray(dy, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, h, wd, co = dy.shape
    kh, kw = w.shape[:2]
    dx = np.empty((n, h, wd, ci), np.float32)
    lib().orc_conv2d_same_bwd_input_f32(_p(dy), _p(w), _p(dx), n, h, wd, ci, co, kh, kw)
    return dx


def conv2d_same_bwd_filter(x, dy, kh, kw, want_bias=False):
    x = np.ascontiguousarray(x, np.float32)
    dy = np.ascontiguousarray(dy, np.float32)
    n, h, wd, ci = x.shape
    co = dy.shape[3]
    dw = np.empty((kh, kw, ci, co), np.float32)
    db = np.empty(co, np.float32) if want_bias else None
    lib().orc_conv2d_same_bwd_filter_f32(_p(x), _p(dy), _p(dw), _p(db), n, h, wd, ci, co, kh, kw)
    return (dw, db) if want_bias else dw


def conv1d_same_fwd(x, w, b=None):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, L, ci = x.shape
    k, _, co = w.shape
    y = np.empty((n, L, co), np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    lib().orc_conv1d_same_fwd_f32(_p(x), _p(w), _p(bb), _p(y), n, L, ci, co, k)
    return y


def lrn_fwd(x, radius=5, bias=1.0, alpha=1.0, beta=0.5):
    x = np.ascontiguousarray(x, np.float32)
    c = x.shape[-1]
    y = np.empty_like(x)
    lib().orc_lrn_fwd_f32(_p(x), _p(y), ctypes.c_size_t(x.size // c), c, radius, ctypes.c_float(bias),
                          ctypes.c_float(alpha), ctypes.c_float(beta))
    return y
