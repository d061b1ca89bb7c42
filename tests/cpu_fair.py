"""ctypes loader of oracle/libsqlrs_cpu_fair.so — the all-core "fair" CPU baseline of the
headline query and the SplitMix64 column generators.  Test / bench
infrastructure only: nothing under sqlrs_amd/ imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
FAIR_SO = os.path.join(ORACLE_DIR, "libsqlrs_cpu_fair.so")

_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "cpu_fair.cpp")
        if (not os.path.exists(FAIR_SO)) or (os.path.exists(src) and os.path.getmtime(FAIR_SO) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        lib = C.CDLL(FAIR_SO)
        i64p, f64p = C.POINTER(C.c_int64), C.POINTER(C.c_double)
        lib.fair_max_threads.restype = C.c_int
        lib.fair_gen_key.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_uint64, i64p]
        lib.fair_gen_val.argtypes = [C.c_uint64, C.c_int64, C.c_int64, f64p]
        lib.fair_gen_dim_key.argtypes = [C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, i64p]
        lib.fair_c5.restype = C.c_int
        lib.fair_c5.argtypes = [C.c_int64, C.c_int64, C.c_double, C.c_int, i64p, f64p, i64p, i64p, i64p, f64p,
                                i64p, f64p]
        _lib = lib
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def max_threads() -> int:
    return load().fair_max_threads()


def gen_c5(n_fact: int, n_dim: int, fact_start: int = 0):
    """(fact_key, fact_val, dim_key) exactly as sqlrs_amd.datagen / bench.py generate them"""
    from sqlrs_amd import datagen
    lib = load()
    fk = np.empty(n_fact, dtype=np.int64)
    fv = np.empty(n_fact, dtype=np.float64)
    dk = np.empty(n_dim, dtype=np.int64)
    lib.fair_gen_key(0xF1, fact_start, n_fact, n_dim, _p(fk, C.c_int64))
    lib.fair_gen_val(0xF2, fact_start, n_fact, _p(fv, C.c_double))
    lib.fair_gen_dim_key(0, n_dim, n_dim, datagen._coprime_multiplier(n_dim), _p(dk, C.c_int64))
    return fk, fv, dk


def run_c5(fact_key, fact_val, dim_key, threshold: float, threads: int = 0):
    """-> (keys, counts, sums, seconds); groups in partition order (sort by key to compare)"""
    lib = load()
    n_dim = len(dim_key)
    ok = np.empty(max(n_dim, 1), dtype=np.int64)
    oc = np.empty(max(n_dim, 1), dtype=np.int64)
    os_ = np.empty(max(n_dim, 1), dtype=np.float64)
    ng, sec = C.c_int64(0), C.c_double(0.0)
    st = lib.fair_c5(len(fact_key), n_dim, threshold, threads, _p(fact_key, C.c_int64), _p(fact_val, C.c_double),
                     _p(dim_key, C.c_int64), _p(ok, C.c_int64), _p(oc, C.c_int64), _p(os_, C.c_double),
                     C.byref(ng), C.byref(sec))
    if st != 0:
        raise MemoryError("fair_c5: allocation failed")
    g = ng.value
    return ok[:g], oc[:g], os_[:g], sec.value
