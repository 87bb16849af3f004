"""Loader of libfandomsearch_hip.so (the C ABI of include/fandom_search.h).

There is no CPU fallback: if the library is missing or cannot be loaded the
import of anything that needs it fails with a clear error.
"""

import ctypes as C
import os
import sys
import subprocess

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libfandomsearch_hip.so"
_LIB = None

SYMBOLS = ("fs_version", "fs_strerror", "fs_last_error", "fs_index_create",
           "fs_index_info_get", "fs_index_destroy", "fs_corpus_create",
           "fs_corpus_destroy", "fs_search_corpus", "fs_search",
           "fs_scan_benchmark", "fs_corpus_update_begin", "fs_corpus_update_end",
           "fs_host_alloc", "fs_host_free", "fs_rows_unpack", "fs_rows_unpack8",
           "fs_reuse_histogram", "fs_reuse_histogram_rows",
           "fs_search_corpus_begin", "fs_search_corpus_end", "fs_index_set_scan_timing",
           "fs_index_reload_switches", "fs_search_kernel_name", "fs_debug_stamps",
           "fs_search_profile", "fs_index_component_sizes", "fs_index_share_info", "fs_index_share_counts", "fs_stream_floor",
           "fs_textenc_create", "fs_textenc_destroy", "fs_textenc_add", "fs_textenc_encode_files",
           "fs_textenc_encode_files_vec",
           "fs_csvw_create", "fs_csvw_destroy", "fs_csvw_set_script", "fs_csvw_add_strings", "fs_csvw_strings",
           "fs_csvw_format")


class FsError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = "%s failed: %d" % (where, code)
        if detail:
            msg += " (%s)" % detail
        RuntimeError.__init__(self, msg)


PRELOAD_TORCH = True    # see load()


def lib_path():
    # FS_LIB_FILE: another build of the same library (A/B measurements of two builds
    # in one GPU session); it must sit next to the regular one
    return os.path.join(_PKG, os.path.basename(os.environ.get("FS_LIB_FILE", LIB_NAME)))


def source_hash():
    """Identifies the kernels a measurement belongs to: SHA-256 (first 16 hex digits) over
    the library's sources (csrc/*.hip, csrc/*.h, include/fandom_search.h), which travel
    with the built .so.  Artifacts under profiles/ carry it, and bench.py refuses a
    traffic figure whose stamp differs from the sources it runs (git is not available
    where the measurements are made)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_PKG, "csrc")
    files = sorted(f for f in os.listdir(src) if f.endswith((".hip", ".h")))
    paths = [os.path.join(src, f) for f in files]
    paths.append(os.path.join(os.path.dirname(_PKG), "include", "fandom_search.h"))
    for path in paths:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles
    without a GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return lib_path()


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built; run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C fandom_search_amd/csrc`. There is no CPU "
            "fallback for the search path." % path)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so
    # (SONAME libamdhip64.so.7).  Loaded first, it satisfies this library's
    # DT_NEEDED libamdhip64.so.7 as well; loaded second, it would come in as a second
    # runtime next to /opt/rocm's (two sets of queues and contexts, and torch can then
    # fail with "No HIP GPUs are available" late in a long process).
    # PRELOAD_TORCH = False (the single-process command line, which never imports torch:
    # its start-up costs a second) loads the library against /opt/rocm's runtime alone.
    if PRELOAD_TORCH or "torch" in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.fs_version.restype = C.c_int
    L.fs_strerror.restype = C.c_char_p
    L.fs_strerror.argtypes = [C.c_int]
    L.fs_last_error.restype = C.c_char_p
    L.fs_index_create.restype = C.c_int
    L.fs_index_create.argtypes = [
        C.POINTER(abi.FsConfig), u32p, u32p, u64p, C.c_uint64,
        C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_double),
        C.POINTER(C.c_void_p)]
    L.fs_index_info_get.restype = C.c_int
    L.fs_index_info_get.argtypes = [C.c_void_p, C.POINTER(abi.FsIndexInfo)]
    L.fs_index_destroy.restype = None
    L.fs_index_destroy.argtypes = [C.c_void_p]
    L.fs_corpus_create.restype = C.c_int
    L.fs_corpus_create.argtypes = [
        C.c_void_p, u32p, u32p, u64p, C.c_uint64, u32p, u64p, C.c_uint64,
        C.POINTER(C.c_void_p)]
    L.fs_corpus_destroy.restype = None
    L.fs_corpus_destroy.argtypes = [C.c_void_p]
    L.fs_search_corpus.restype = C.c_int
    L.fs_search_corpus.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, u64p,
        C.POINTER(abi.FsStats)]
    L.fs_search.restype = C.c_int
    L.fs_search.argtypes = [
        C.c_void_p, u32p, u32p, u64p, C.c_uint64, u32p, u64p, C.c_uint64,
        C.c_void_p, C.c_uint64, u64p, C.POINTER(abi.FsStats)]
    L.fs_corpus_update_begin.restype = C.c_int
    L.fs_corpus_update_begin.argtypes = [C.c_void_p, u32p, u32p, u64p, C.c_uint64]
    L.fs_corpus_update_end.restype = C.c_int
    L.fs_corpus_update_end.argtypes = [C.c_void_p]
    L.fs_host_alloc.restype = C.c_int
    L.fs_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
    L.fs_host_free.restype = None
    L.fs_host_free.argtypes = [C.c_void_p]
    L.fs_rows_unpack.restype = C.c_int
    L.fs_rows_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.fs_rows_unpack8.restype = C.c_int
    L.fs_rows_unpack8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                  C.c_void_p]
    L.fs_reuse_histogram.restype = C.c_int
    L.fs_reuse_histogram.argtypes = [C.c_int, u32p, C.POINTER(C.c_double), C.c_uint64, C.c_uint64,
                                     C.POINTER(C.c_double), C.c_uint32, u32p]
    L.fs_reuse_histogram_rows.restype = C.c_int
    L.fs_reuse_histogram_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.POINTER(C.c_double), C.c_uint32, C.c_void_p]
    L.fs_search_corpus_begin.restype = C.c_int
    L.fs_search_corpus_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                         C.POINTER(C.c_uint32)]
    L.fs_search_corpus_end.restype = C.c_int
    L.fs_search_corpus_end.argtypes = [C.c_void_p, C.c_uint32, u64p, C.POINTER(abi.FsStats)]
    L.fs_index_set_scan_timing.restype = C.c_int
    L.fs_index_set_scan_timing.argtypes = [C.c_void_p, C.c_uint32]
    L.fs_search_kernel_name.restype = C.c_char_p
    L.fs_search_kernel_name.argtypes = [C.c_void_p, C.c_void_p]
    L.fs_index_reload_switches.restype = C.c_int
    L.fs_index_reload_switches.argtypes = [C.c_void_p]
    if hasattr(L, "fs_debug_stamps"):      # (absent from older builds loaded through FS_LIB_FILE)
        L.fs_debug_stamps.restype = C.c_int
        L.fs_debug_stamps.argtypes = [C.c_void_p, C.c_uint32, u64p, C.c_uint64, u64p]
    if hasattr(L, "fs_stream_floor"):
        L.fs_stream_floor.restype = C.c_int
        L.fs_stream_floor.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
    if hasattr(L, "fs_index_share_counts"):
        L.fs_index_share_counts.restype = C.c_int
        L.fs_index_share_counts.argtypes = [C.c_void_p, u64p]
    if hasattr(L, "fs_index_share_info"):
        L.fs_index_share_info.restype = C.c_int
        L.fs_index_share_info.argtypes = [C.c_void_p, u32p, u32p, u32p, C.POINTER(C.c_double)]
    if hasattr(L, "fs_index_component_sizes"):
        L.fs_index_component_sizes.restype = C.c_int
        L.fs_index_component_sizes.argtypes = [C.c_void_p, u32p, C.c_uint64, u64p, u32p]
    if hasattr(L, "fs_search_profile"):    # (absent from older builds loaded through FS_LIB_FILE)
        L.fs_search_profile.restype = C.c_int
        L.fs_search_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                        C.c_char_p, C.c_uint64, C.POINTER(C.c_double), C.c_uint32,
                                        C.POINTER(C.c_uint32)]
    L.fs_scan_benchmark.restype = C.c_int
    L.fs_scan_benchmark.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32,
                                    C.POINTER(C.c_double)]
    if L.fs_version() != 1:
        raise RuntimeError("ABI version mismatch in %s" % path)
    _LIB = L
    return L


def check(rc, where):
    if rc != abi.FS_OK:
        L = load()
        detail = L.fs_strerror(rc).decode()
        last = L.fs_last_error().decode()
        if last:
            detail += ": " + last
        raise FsError(rc, where, detail)
