"""Native batch files: fs_row records -> the bytes of the reference's CSV (fs_csvw_*, csrc/fs_csv.hip).

The reference makes a record's twelve fields in Python (/root/reference/search.py:192-218)
and writes a batch with csv.writer (:331-334); search.join_records / write_records restate
that, and stay the oracle of this path (tests/test_csvw.py compares bytes).  Here the join
and the formatting are one native call per batch: the script's four columns and the
vocabulary's strings live in the writer (the latter grow with the vocabulary), a batch is its
fs_row array, the works' names and one string id per record.

FANDOM_SEARCH_NATIVE_CSV=0 switches it off (the batch files are then written by the forked
workers of the token pool, as before)."""

import ctypes as C
import os
import threading

import numpy as np

from . import _lib, abi


def enabled():
    return os.environ.get("FANDOM_SEARCH_NATIVE_CSV", "1") != "0"


def _bind(L):
    if getattr(L, "_csvw_bound", False):
        return L
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.fs_csvw_create.restype = C.c_int
    L.fs_csvw_create.argtypes = [C.POINTER(C.c_void_p)]
    L.fs_csvw_destroy.restype = None
    L.fs_csvw_destroy.argtypes = [C.c_void_p]
    L.fs_csvw_set_script.restype = C.c_int
    L.fs_csvw_set_script.argtypes = [C.c_void_p, C.c_uint64] + [C.c_char_p, u64p] * 4
    L.fs_csvw_add_strings.restype = C.c_int
    L.fs_csvw_add_strings.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_uint64]
    L.fs_csvw_strings.restype = C.c_uint64
    L.fs_csvw_strings.argtypes = [C.c_void_p]
    L.fs_csvw_format.restype = C.c_int
    L.fs_csvw_format.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, u64p, C.c_uint64, u32p,
                                 C.POINTER(u8p), u64p]
    L._csvw_bound = True
    return L


def _table(texts):
    """(bytes, offsets) of a list of str; None is the empty string (csv.writer writes it so)."""
    raw = [b"" if t is None else str(t).encode("utf-8") for t in texts]
    off = np.zeros(len(raw) + 1, dtype=np.uint64)
    if raw:
        off[1:] = np.cumsum([len(b) for b in raw], dtype=np.uint64)
    return b"".join(raw), off


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class CsvWriter(object):
    """format(filenames, rows, fan_sids) -> bytes of the batch file;
    write_async(path, ...) formats and writes on a background thread (the GIL is released
    inside the library and inside the file write), finish() waits and raises what a write raised."""

    def __init__(self, word_lowercase, orth_id, character, scene, strings, writers=0):
        self._L = _bind(_lib.load())
        tabs = [_table(word_lowercase), _table(orth_id), _table(character), _table(scene)]
        args = []
        for b, off in tabs:
            args += [b, _u64p(off)]
        # a native writer formats one batch at a time (its tables and its output buffer are its
        # own): `writers` of them take the batches in turn, so that several batches are formatted
        # at once (a batch of 15 000 records takes one thread 3 ms; batches come every 2 ms)
        if not writers:
            from .search import usable_cpus
            writers = max(1, min(4, usable_cpus() // 4))
        self._hs = []
        for _ in range(int(writers)):
            h = C.c_void_p()
            _lib.check(self._L.fs_csvw_create(C.byref(h)), "fs_csvw_create")
            self._hs.append((h, threading.Lock()))
            _lib.check(self._L.fs_csvw_set_script(h, len(word_lowercase), *args), "fs_csvw_set_script")
        self.strings = strings            # the vocabulary's list (grows)
        self._turn = 0
        self._jobs = []

    def close(self):
        hs, self._hs = self._hs, []
        for h, _ in hs:
            self._L.fs_csvw_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _sync_strings(self, h):
        have = int(self._L.fs_csvw_strings(h))
        n = len(self.strings)
        if have < n:
            b, off = _table(self.strings[have:n])
            _lib.check(self._L.fs_csvw_add_strings(h, b, _u64p(off), n - have), "fs_csvw_add_strings")

    def format(self, filenames, rows, fan_sids, turn=0):
        rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
        sids = np.ascontiguousarray(fan_sids, dtype=np.uint32)
        if len(sids) != len(rows):
            raise ValueError("one string id per record")
        names, noff = _table(filenames)
        out, n = C.POINTER(C.c_uint8)(), C.c_uint64()
        h, lock = self._hs[turn % len(self._hs)]
        with lock:
            self._sync_strings(h)
            _lib.check(self._L.fs_csvw_format(
                h, rows.ctypes.data_as(C.c_void_p), len(rows), names, _u64p(noff), len(filenames),
                sids.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(out), C.byref(n)), "fs_csvw_format")
            return C.string_at(out, n.value)

    def write(self, path, filenames, rows, fan_sids, turn=0):
        data = self.format(filenames, rows, fan_sids, turn)
        with open(path, "wb") as fh:
            fh.write(data)
        return len(rows)

    def write_async(self, path, filenames, rows, fan_sids):
        box = {}
        args = (path, list(filenames), np.array(rows, dtype=abi.ROW_DTYPE, copy=True),
                np.array(fan_sids, dtype=np.uint32, copy=True), self._turn)
        self._turn += 1

        def run():
            try:
                box["n"] = self.write(*args)
            except BaseException as e:          # (raised again by finish())
                box["e"] = e

        th = threading.Thread(target=run, daemon=True)
        th.start()
        self._jobs.append((th, box))
        while len(self._jobs) > 8:               # bounded: the oldest first
            self._join(self._jobs.pop(0))

    @staticmethod
    def _join(job):
        job[0].join()
        if "e" in job[1]:
            raise job[1]["e"]

    def finish(self):
        jobs, self._jobs = self._jobs, []
        err = None
        for job in jobs:
            try:
                self._join(job)
            except BaseException as e:
                err = err or e
        if err is not None:
            raise err
