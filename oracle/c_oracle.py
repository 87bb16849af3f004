"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/liboracle.so
(oracle/fs_oracle.c, the plain-C restatement of the reference's search path).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package.
"""

import ctypes as C
import os
import subprocess

import numpy as np

from fandom_search_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return os.path.join(_HERE, "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.fo_index_create.restype = C.c_int
        L.fo_index_create.argtypes = [
            C.POINTER(abi.FsConfig), u32p, u32p, u64p, C.c_uint64,
            C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_double), C.c_int,
            C.POINTER(C.c_void_p)]
        L.fo_index_destroy.restype = None
        L.fo_index_destroy.argtypes = [C.c_void_p]
        L.fo_search.restype = C.c_int
        L.fo_search.argtypes = [
            C.c_void_p, u32p, u32p, u64p, C.c_uint64, u32p, u64p, C.c_uint64,
            C.c_void_p, C.c_uint64, u64p, C.POINTER(abi.FsStats)]
        L.fo_script_keys.restype = C.c_int
        L.fo_script_keys.argtypes = [C.c_void_p, C.c_uint64, u32p]
        _LIB = L
    return _LIB


class OracleIndex(object):
    """Reference-algorithm search on the CPU (canonical arithmetic)."""

    def __init__(self, cfg, script_vec, script_chars, script_off, emb,
                 normals, threads=1):
        self.cfg = cfg
        self._keep = (abi.as_u32(script_vec), abi.as_u32(script_chars),
                      abi.as_u64(script_off),
                      np.ascontiguousarray(emb, dtype=np.float32),
                      np.ascontiguousarray(normals, dtype=np.float64))
        sv, sc, so, e, nm = self._keep
        if nm.size != cfg.number_of_hashes * cfg.hash_dimensions \
                * cfg.emb_dim * cfg.window_size:
            raise ValueError("normals shape")
        self._h = C.c_void_p()
        rc = lib().fo_index_create(
            C.byref(cfg), abi.ptr(sv, C.c_uint32), abi.ptr(sc, C.c_uint32),
            abi.ptr(so, C.c_uint64), len(sv), abi.ptr(e, C.c_float),
            e.shape[0], abi.ptr(nm, C.c_double), threads, C.byref(self._h))
        if rc != 0:
            raise RuntimeError("fo_index_create -> %d" % rc)

    def search(self, tok_vec, work_off, str_chars, str_off, tok_str=None):
        tv = abi.as_u32(tok_vec)
        ts = abi.as_u32(tok_str) if tok_str is not None else None
        wo = abi.as_u64(work_off)
        sc = abi.as_u32(str_chars)
        so = abi.as_u64(str_off)
        st = abi.FsStats()
        n = C.c_uint64(0)
        cap = 1 << 16
        while True:
            rows = np.empty(cap, dtype=abi.ROW_DTYPE)
            rc = lib().fo_search(
                self._h, abi.ptr(tv, C.c_uint32), abi.ptr(ts, C.c_uint32),
                abi.ptr(wo, C.c_uint64), len(wo) - 1, abi.ptr(sc, C.c_uint32),
                abi.ptr(so, C.c_uint64), len(so) - 1,
                rows.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(st))
            if rc == abi.FS_E_CAPACITY:
                cap = int(n.value)
                continue
            if rc != 0:
                raise RuntimeError("fo_search -> %d" % rc)
            return rows[:n.value].copy(), st

    def script_keys(self, window):
        keys = np.zeros(self.cfg.number_of_hashes, dtype=np.uint32)
        rc = lib().fo_script_keys(self._h, window, abi.ptr(keys, C.c_uint32))
        if rc != 0:
            raise RuntimeError("fo_script_keys -> %d" % rc)
        return keys

    def close(self):
        if self._h:
            lib().fo_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
