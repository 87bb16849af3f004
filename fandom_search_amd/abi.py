"""ctypes mirrors of include/fandom_search.h (structs, constants, dtypes)."""

import ctypes as C

import numpy as np

FS_OK = 0
FS_E_INVALID = -1
FS_E_NOMEM = -2
FS_E_DEVICE = -3
FS_E_CAPACITY = -4
FS_E_UNSUPPORTED = -5
FS_E_UNPROVEN = -6

FS_MODE_AUTO = 0
FS_MODE_GENERAL = 1
FS_MODE_EXACT = 2

FS_OOV_FLAG = 0x80000000

FS_ROWS_HOST = 0
FS_ROWS_DEVICE = 1
FS_ROWS_DEVICE_PACKED = 2
FS_ROWS_DEVICE_PACKED8 = 3
FS_ROWS_HEADER = 0x100
ROWS_HEADER_BYTES = 32
PACKED_ROW_BYTES = 16
PACKED8_ROW_BYTES = 8
PACKED8_MAX_SCRIPT = 1 << 18


class FsConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("window_size", C.c_uint32),
                ("number_of_hashes", C.c_uint32),
                ("hash_dimensions", C.c_uint32),
                ("emb_dim", C.c_uint32),
                ("nearest_n", C.c_uint32),
                ("unique_filter", C.c_uint32),
                ("mode", C.c_uint32),
                ("device", C.c_int32),
                ("reserved", C.c_uint32),
                ("distance_threshold", C.c_double)]


class FsStats(C.Structure):
    _fields_ = [("windows_processed", C.c_uint64),
                ("candidates", C.c_uint64),
                ("matches", C.c_uint64),
                ("rows", C.c_uint64),
                ("scan_ms", C.c_double),
                ("total_ms", C.c_double),
                ("path", C.c_uint32),
                ("scan_launches", C.c_uint32),
                ("lsh_pending", C.c_uint32),
                ("handoff_fallbacks", C.c_uint32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class FsIndexInfo(C.Structure):
    _fields_ = [("path", C.c_uint32),
                ("proof_ok", C.c_uint32),
                ("c_max", C.c_double),
                ("cos_bound", C.c_double),
                ("norm_min", C.c_double),
                ("norm_max", C.c_double),
                ("n_script", C.c_uint64),
                ("n_windows", C.c_uint64),
                ("n_grams", C.c_uint64),
                ("filter_bytes", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# fs_row: 32 bytes
ROW_DTYPE = np.dtype([("work", np.uint32), ("fan_ix", np.uint32),
                      ("orig_ix", np.uint32), ("lev", np.uint32),
                      ("dist", np.float64), ("comb", np.float64)])
assert ROW_DTYPE.itemsize == 32


def default_unique_filter():
    """Whether a query's bucket contents go through NearPy's UniqueFilter before the
    distances are taken.  OFF by default: the reference calls `engine.neighbours(row)`
    with no arguments (/root/reference/search.py:178), and NearPy 1.0.0 -- the release pip
    installed when the reference was written; requirements.txt pins none -- applies fetch
    filters only when they are passed to neighbours() itself (`if fetch_vector_filters:`
    on the ARGUMENT; the engine's own default [UniqueFilter()] is never consulted there),
    so a script window found under k of the 15 hashes comes back k times and takes k of
    NearestFilter(10)'s places.  NearPy 0.2.x applied the engine's filter:
    FANDOM_SEARCH_UNIQUE_FILTER=1 (or `ao3.py search --unique-filter 1`) gives that."""
    import os
    return os.environ.get("FANDOM_SEARCH_UNIQUE_FILTER", "0").strip() not in ("", "0", "false", "no", "off")


def make_config(window_size=6, number_of_hashes=15, hash_dimensions=14,
                distance_threshold=0.1, emb_dim=300, nearest_n=10,
                unique_filter=None, mode=FS_MODE_AUTO, device=0):
    """Defaults are the keyword defaults of the reference's analyze()
    (/root/reference/search.py:336-341) and what NearPy's Engine does with the defaults
    the reference leaves it (unique_filter: default_unique_filter())."""
    if unique_filter is None:
        unique_filter = default_unique_filter()
    cfg = FsConfig()
    cfg.struct_size = C.sizeof(FsConfig)
    cfg.window_size = window_size
    cfg.number_of_hashes = number_of_hashes
    cfg.hash_dimensions = hash_dimensions
    cfg.emb_dim = emb_dim
    cfg.nearest_n = nearest_n
    cfg.unique_filter = 1 if unique_filter else 0
    cfg.mode = mode
    cfg.device = device
    cfg.distance_threshold = distance_threshold
    return cfg


def ptr(arr, ctype):
    """Typed pointer to a C-contiguous numpy array (None -> NULL)."""
    if arr is None:
        return None
    return arr.ctypes.data_as(C.POINTER(ctype))


def as_u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def as_u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)
