"""ctypes loader for oracle/bear_oracle.c (TEST INFRASTRUCTURE ONLY; see bear_oracle.py header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbear_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        f64p = ctypes.POINTER(ctypes.c_double)
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_dm_prior_f64.argtypes = [u32p, f64p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_int, f64p, f64p, ctypes.c_int]
        L.oracle_dm_prior_f64.restype = None
        L.oracle_dm_ref_f64.argtypes = [u32p, u32p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_int, f64p, ctypes.c_int]
        L.oracle_dm_ref_f64.restype = None
        L.oracle_dm_prior_mass_f64.argtypes = [u32p, f64p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, f64p, ctypes.c_int]
        L.oracle_dm_prior_mass_f64.restype = None
        L.oracle_dm_ref_mass_f64.argtypes = [u32p, u32p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_int, f64p, ctypes.c_int]
        L.oracle_dm_ref_mass_f64.restype = None
        _lib = L
    return _lib


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def max_threads():
    return lib().oracle_max_threads()


def dm_prior(counts, prior, h_signed, eps=1e-7, train_ar=False, want_grad=False, nthreads=1):
    c, cp = _u32(counts)
    f, fp = _f64(prior)
    assert c.shape == f.shape and c.shape[-1] == 5
    out, op = _f64(np.zeros(2))
    if want_grad:
        g, gp = _f64(np.zeros(c.shape))
    else:
        g, gp = None, None
    lib().oracle_dm_prior_f64(cp, fp, c.shape[0], h_signed, eps, int(train_ar), op, gp, nthreads)
    return out, g


def dm_ref(train, ref, h_signed, tau_signed, nu_signed, eps=1e-7, train_ar=False, nthreads=1):
    c, cp = _u32(train)
    r, rp = _u32(ref)
    assert c.shape == r.shape and c.shape[-1] == 5
    out, op = _f64(np.zeros(4))
    lib().oracle_dm_ref_f64(cp, rp, c.shape[0], h_signed, tau_signed, nu_signed, eps, int(train_ar), op, nthreads)
    return out


def dm_prior_mass(counts, prior, h_signed, eps=1e-7, nthreads=1):
    """L1 mass of d sum LL / d h_signed (BEAR mode): the sum of the absolute values of its per-row-and-letter terms -- the scale
    the parity tests bound that gradient's error by."""
    c, cp = _u32(counts)
    f, fp = _f64(prior)
    out, op = _f64(np.zeros(1))
    lib().oracle_dm_prior_mass_f64(cp, fp, c.shape[0], h_signed, eps, op, nthreads)
    return float(out[0])


def dm_ref_mass(train, ref, h_signed, tau_signed, nu_signed, eps=1e-7, train_ar=False, nthreads=1):
    """L1 masses [d/dh_signed, d/dtau_signed, d/dnet_weight_signed] of the bear_ref gradients."""
    c, cp = _u32(train)
    r, rp = _u32(ref)
    out, op = _f64(np.zeros(3))
    lib().oracle_dm_ref_mass_f64(cp, rp, c.shape[0], h_signed, tau_signed, nu_signed, eps, int(train_ar), op, nthreads)
    return out
