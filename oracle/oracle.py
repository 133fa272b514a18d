"""ctypes loader for the CPU oracle (oracle/ntt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_p64 = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "ntt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.oracle_mulmod.restype = _u64
        L.oracle_mulmod.argtypes = [_u64, _u64, _u64]
        L.oracle_powmod.restype = _u64
        L.oracle_powmod.argtypes = [_u64, _u64, _u64]
        L.oracle_invmod.restype = _u64
        L.oracle_invmod.argtypes = [_u64, _u64]
        L.oracle_bitrev.restype = _u32
        L.oracle_bitrev.argtypes = [_u32, ctypes.c_int]
        L.oracle_is_prime.restype = ctypes.c_int
        L.oracle_is_prime.argtypes = [_u64]
        L.oracle_find_prime.restype = _u64
        L.oracle_find_prime.argtypes = [ctypes.c_int, _u32, ctypes.c_int]
        L.oracle_min_root.restype = _u64
        L.oracle_min_root.argtypes = [_u64, _u32]
        L.oracle_make_tables.restype = None
        L.oracle_make_tables.argtypes = [_u64, _u64, _u32, _p64, _p64]
        L.oracle_make_inv_tables.restype = None
        L.oracle_make_inv_tables.argtypes = [_u64, _u64, _u32, _p64, _p64]
        L.oracle_forward.restype = None
        L.oracle_forward.argtypes = [_p64, _p64, _u64, _p64, _p64, _p64, _u32, _u64]
        L.oracle_naive_forward.restype = None
        L.oracle_naive_forward.argtypes = [_p64, _u64, _u64, _p64, _u32]
        L.oracle_inverse.restype = None
        L.oracle_inverse.argtypes = [_p64, _u64, _p64, _p64, _u32, _u64]
        L.oracle_pointwise.restype = None
        L.oracle_pointwise.argtypes = [_p64, _p64, _u64, _p64, _u64]
        L.oracle_negacyclic_schoolbook.restype = None
        L.oracle_negacyclic_schoolbook.argtypes = [_p64, _p64, _u64, _p64, _u32]
        L.oracle_fill_splitmix.restype = None
        L.oracle_fill_splitmix.argtypes = [_p64, _u64, _u64, _u64]
        L.oracle_fnv1a_words.restype = _u64
        L.oracle_fnv1a_words.argtypes = [_p64, _u64]
        L.oracle_forward_mt.restype = ctypes.c_int
        L.oracle_forward_mt.argtypes = [_p64, _u64, _p64, _p64, _p64, _u32, _u64, ctypes.c_int]
        _lib = L
    return _lib


def _ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p64)


def find_prime(bits, n, k=0):
    return int(lib().oracle_find_prime(bits, n, k))


def min_root(q, n):
    return int(lib().oracle_min_root(q, n))


def is_prime(q):
    return bool(lib().oracle_is_prime(q))


def make_tables(q, psi, n):
    tw = np.empty(n, dtype=np.uint64)
    pre = np.empty(n, dtype=np.uint64)
    lib().oracle_make_tables(q, psi, n, _ptr(tw), _ptr(pre))
    return tw, pre


def make_inv_tables(q, psi, n):
    tw = np.empty(n, dtype=np.uint64)
    pre = np.empty(n, dtype=np.uint64)
    lib().oracle_make_inv_tables(q, psi, n, _ptr(tw), _ptr(pre))
    return tw, pre


def forward(x, q, tw, pre, n, x2=None):
    """x: (..., n) uint64 frames under one modulus -> forward NTT (reference path)."""
    x = np.ascontiguousarray(x, dtype=np.uint64)
    x2 = x if x2 is None else np.ascontiguousarray(x2, dtype=np.uint64)
    out = np.empty_like(x)
    frames = x.size // n
    lib().oracle_forward(_ptr(x), _ptr(x2), q, _ptr(tw), _ptr(pre), _ptr(out), n, frames)
    return out


def forward_mt(x, q, tw, pre, n, threads):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.empty_like(x)
    rc = lib().oracle_forward_mt(_ptr(x), q, _ptr(tw), _ptr(pre), _ptr(out), n, x.size // n, threads)
    assert rc == 0
    return out


def naive_forward(x, q, psi, n):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.empty(n, dtype=np.uint64)
    lib().oracle_naive_forward(_ptr(x), q, psi, _ptr(out), n)
    return out


def inverse(y, q, itw, n):
    y = np.ascontiguousarray(y, dtype=np.uint64)
    out = np.empty_like(y)
    lib().oracle_inverse(_ptr(y), q, _ptr(itw), _ptr(out), n, y.size // n)
    return out


def pointwise(a, b, q):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    c = np.empty_like(a)
    lib().oracle_pointwise(_ptr(a), _ptr(b), q, _ptr(c), a.size)
    return c


def schoolbook(a, b, q, n):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    c = np.empty(n, dtype=np.uint64)
    lib().oracle_negacyclic_schoolbook(_ptr(a), _ptr(b), q, _ptr(c), n)
    return c


def fill_splitmix(count, seed, q):
    x = np.empty(count, dtype=np.uint64)
    lib().oracle_fill_splitmix(_ptr(x), count, seed, q)
    return x


def fnv1a_words(x):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    return int(lib().oracle_fnv1a_words(_ptr(x), x.size))
