"""Python face of the C ABI: script index, device-resident corpus, search.

`ScriptIndex` stands where AnnIndexSearch.__init__ / build_lsh_engine stand in
the reference (/root/reference/search.py:131-154, 86-124); `Corpus` is a batch
of tokenised works already in HBM; `ScriptIndex.search` is
AnnIndexSearch.search (search.py:163-226) over the whole batch and returns the
numeric half of the records (abi.ROW_DTYPE), sorted by (work, fan word index).
"""

import ctypes as C
import weakref

import numpy as np

import atexit

from . import _lib, abi
from .vocab import pack_strings

# Handles still open when the interpreter exits are closed before the HIP runtime's
# own teardown (a __del__ that runs after it would call into a runtime that is gone).
_LIVE = weakref.WeakSet()


@atexit.register
def _close_all():
    for obj in sorted(_LIVE, key=lambda o: 0 if isinstance(o, ScriptIndex) else 1):
        try:
            obj.close()            # an index closes its corpora first
        except Exception:
            pass



class Corpus(object):
    def __init__(self, index, tok_vec, work_off, str_chars, str_off,
                 tok_str=None):
        self.index = index
        self.tok_vec = abi.as_u32(tok_vec)
        self.tok_str = abi.as_u32(tok_str) if tok_str is not None else None
        self.work_off = abi.as_u64(work_off)
        chars = abi.as_u32(str_chars)
        off = abi.as_u64(str_off)
        self.n_works = len(self.work_off) - 1
        self.n_tok = int(self.work_off[-1])
        if self.tok_vec.size < self.n_tok:
            raise ValueError("token buffer shorter than work_off[-1]")
        self._h = C.c_void_p()
        L = _lib.load()
        _lib.check(L.fs_corpus_create(
            index._h, abi.ptr(self.tok_vec, C.c_uint32),
            abi.ptr(self.tok_str, C.c_uint32),
            abi.ptr(self.work_off, C.c_uint64), self.n_works,
            abi.ptr(chars, C.c_uint32), abi.ptr(off, C.c_uint64),
            len(off) - 1, C.byref(self._h)), "fs_corpus_create")
        index._corpora.add(self)
        _LIVE.add(self)

    def update_begin(self, tok_vec, work_off, tok_str=None):
        """Queue the upload of a new batch of works into this corpus (copy
        stream; returns at once).  The arrays must stay alive and untouched until
        update_end() or the next search on this corpus; pass pinned arrays
        (`pinned_array`) for the copy to overlap a running search."""
        self.tok_vec = abi.as_u32(tok_vec)
        self.tok_str = abi.as_u32(tok_str) if tok_str is not None else None
        self.work_off = abi.as_u64(work_off)
        self.n_works = len(self.work_off) - 1
        self.n_tok = int(self.work_off[-1])
        _lib.check(_lib.load().fs_corpus_update_begin(
            self._h, abi.ptr(self.tok_vec, C.c_uint32), abi.ptr(self.tok_str, C.c_uint32),
            abi.ptr(self.work_off, C.c_uint64), self.n_works), "fs_corpus_update_begin")

    def update_end(self):
        _lib.check(_lib.load().fs_corpus_update_end(self._h), "fs_corpus_update_end")

    def close(self):
        if self._h:
            _lib.load().fs_corpus_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def torch_ready():
    """The library launches on streams of its own (hipStreamNonBlocking: no implicit ordering
    with torch's stream).  Device memory that torch has only just produced -- a zero-filled
    buffer, a host-to-device copy, a clone -- must be complete before it is handed to a call
    that reads it or writes into it: otherwise the library's kernel can read what the allocator
    left there from an earlier batch, or torch's fill can land on top of the library's records.
    (Round 5: `unpack_shard` read a stale copy of a shard's work offsets once in five runs of the
    two-rank test and attributed a run of records to the work in front.)  Not for per-step paths:
    buffers made ahead of time need none of this."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.current_stream().synchronize()


class PinnedBuffer(object):
    """Page-locked host memory (hipHostMalloc) viewed as a numpy array."""

    def __init__(self, count, dtype):
        self.dtype = np.dtype(dtype)
        self._p = C.c_void_p()
        nbytes = max(1, int(count)) * self.dtype.itemsize
        _lib.check(_lib.load().fs_host_alloc(nbytes, C.byref(self._p)), "fs_host_alloc")
        buf = (C.c_char * nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(count))

    def close(self):
        if self._p:
            self.array = None
            _lib.load().fs_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search_stream(index, batches, str_chars, str_off):
    """Search a corpus that arrives batch by batch (BASELINE configs[4]).

    `batches` yields (tok_vec, work_off) pairs, or (tok_vec, work_off, tok_str).
    Two device corpora alternate: while batch i is searched, batch i+1 is copied
    to the GPU on the other corpus's copy stream.  Yields (rows, stats) per batch,
    work indices local to the batch; the rows are a view of a buffer that the next batch
    overwrites (ScriptIndex.search(reuse=True))."""
    it = iter(batches)
    slots = [None, None]
    cur = None

    def stage(slot, batch):
        tok_str = batch[2] if len(batch) > 2 else None
        if slots[slot] is None:
            slots[slot] = index.corpus(batch[0], batch[1], str_chars, str_off, tok_str=tok_str)
        else:
            slots[slot].update_begin(batch[0], batch[1], tok_str=tok_str)

    try:
        first = next(it)
    except StopIteration:
        return
    stage(0, first)
    cur = 0
    while cur is not None:
        nxt_batch = next(it, None)
        nxt = None
        if nxt_batch is not None:
            nxt = cur ^ 1
            stage(nxt, nxt_batch)            # queued; overlaps the search below
        yield index.search(slots[cur], reuse=True)
        cur = nxt
    for c in slots:
        if c is not None:
            c.close()


class ScriptIndex(object):
    def __init__(self, script_vec, script_words, emb, normals, cfg=None,
                 **cfg_kw):
        L = _lib.load()
        self.cfg = cfg or abi.make_config(**cfg_kw)
        sv = abi.as_u32(script_vec)
        chars, off = pack_strings(list(script_words))
        if len(off) - 1 != len(sv):
            raise ValueError("one word per script token expected")
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        if emb.ndim != 2 or emb.shape[1] != self.cfg.emb_dim:
            raise ValueError("embedding must be (V, %d)" % self.cfg.emb_dim)
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        want = (self.cfg.number_of_hashes * self.cfg.hash_dimensions
                * self.cfg.emb_dim * self.cfg.window_size)
        if normals.size != want:
            raise ValueError("normals must hold H*B*D*n = %d values" % want)
        self._corpora = weakref.WeakSet()      # live Corpus objects of this index
        _LIVE.add(self)
        self._h = C.c_void_p()
        _lib.check(L.fs_index_create(
            C.byref(self.cfg), abi.ptr(sv, C.c_uint32),
            abi.ptr(chars, C.c_uint32), abi.ptr(off, C.c_uint64), len(sv),
            abi.ptr(emb, C.c_float), emb.shape[0],
            abi.ptr(normals, C.c_double), C.byref(self._h)),
            "fs_index_create")
        info = abi.FsIndexInfo()
        _lib.check(L.fs_index_info_get(self._h, C.byref(info)),
                   "fs_index_info_get")
        self.info = info.as_dict()

    def corpus(self, tok_vec, work_off, str_chars, str_off, tok_str=None):
        return Corpus(self, tok_vec, work_off, str_chars, str_off, tok_str)

    def search(self, corpus, cap=None, reuse=False):
        """Rows (numpy structured array, host) and stats of one batch.  `reuse`: the rows
        are a view of a buffer the index keeps and writes again on its next search with
        `reuse` (a fresh 40 MB array per call costs more in page faults than the copy over
        PCIe); take a copy of what must outlive that."""
        L = _lib.load()
        st = abi.FsStats()
        n = C.c_uint64(0)
        cap = int(cap) if cap else max(1024, corpus.n_tok // 16)
        while True:
            if reuse:
                if getattr(self, "_rows_buf", None) is None or len(self._rows_buf) < cap:
                    self._rows_buf = np.empty(cap, dtype=abi.ROW_DTYPE)
                rows = self._rows_buf
                cap = len(rows)
            else:
                rows = np.empty(cap, dtype=abi.ROW_DTYPE)
            rc = L.fs_search_corpus(self._h, corpus._h,
                                    rows.ctypes.data_as(C.c_void_p), cap, 0,
                                    C.byref(n), C.byref(st))
            if rc == abi.FS_E_CAPACITY:
                cap = int(n.value)
                continue
            _lib.check(rc, "fs_search_corpus")
            return rows[:n.value], st

    @staticmethod
    def _rows_mode(packed):
        """packed: False (32-byte fs_row), True or 16 (16-byte wire records), 8
        (8-byte wire records)."""
        if packed == 8:
            return abi.FS_ROWS_DEVICE_PACKED8
        return abi.FS_ROWS_DEVICE_PACKED if packed else abi.FS_ROWS_DEVICE

    def search_device(self, corpus, rows_ptr, cap, packed=False):
        """Rows written to a caller-owned device buffer (`rows_ptr`: 16-byte
        aligned address on this index's device, `cap` records of 32 bytes, or of
        16 / 8 bytes when `packed` is True / 8: the wire formats of the exact
        pipeline, see unpack_device / unpack8_device).  Returns (n_rows, stats);
        raises FsError(FS_E_CAPACITY) with .required when the buffer is too small."""
        L = _lib.load()
        st = abi.FsStats()
        n = C.c_uint64(0)
        mode = self._rows_mode(packed)
        rc = L.fs_search_corpus(self._h, corpus._h, C.c_void_p(rows_ptr),
                                int(cap), mode, C.byref(n), C.byref(st))
        if rc == abi.FS_E_CAPACITY:
            err = _lib.FsError(rc, "fs_search_corpus", "row buffer too small")
            err.required = int(n.value)
            raise err
        _lib.check(rc, "fs_search_corpus")
        return int(n.value), st

    def set_scan_timing(self, period):
        """Attach timing events to every `period`-th scan launch only."""
        _lib.check(_lib.load().fs_index_set_scan_timing(self._h, int(period)),
                   "fs_index_set_scan_timing")

    def kernel_name(self, corpus):
        """Diagnostics: the kernel that dominates a search of `corpus` (profile name)."""
        return _lib.load().fs_search_kernel_name(self._h, corpus._h).decode()

    def stream_floor(self, corpus, reps=20):
        """Diagnostics: ms of a kernel that only reads `corpus`' ids in k_scan_rows' launch
        shape (fs_stream_floor)."""
        ms = C.c_double()
        _lib.check(_lib.load().fs_stream_floor(self._h, corpus._h, reps, C.byref(ms)), "fs_stream_floor")
        return ms.value

    def component_sizes(self):
        """Diagnostics: (sizes of the components of near vectors, whether the prefilters use
        them) -- fs_index_component_sizes."""
        n, used = C.c_uint64(), C.c_uint32()
        L = _lib.load()
        _lib.check(L.fs_index_component_sizes(self._h, None, 0, C.byref(n), C.byref(used)), "fs_index_component_sizes")
        sizes = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            _lib.check(L.fs_index_component_sizes(self._h, abi.ptr(sizes, C.c_uint32), n.value, C.byref(n),
                                                  C.byref(used)), "fs_index_component_sizes")
        return sizes, bool(used.value)

    def share_info(self):
        """Diagnostics: the share rule of the LSH pipeline on this index (fs_index_share_info):
        {"flags", "components", "largest", "gamma"}; flags 0 = not in use."""
        f, n, m, g = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_double()
        _lib.check(_lib.load().fs_index_share_info(self._h, C.byref(f), C.byref(n), C.byref(m), C.byref(g)),
                   "fs_index_share_info")
        return {"flags": f.value, "components": n.value, "largest": m.value, "gamma": g.value}

    def share_counts(self):
        """Diagnostics (FS_SHARE_COUNT=1 when the index was built): what passed what in k_share_scan
        since the last call -- fs_index_share_counts."""
        out = (C.c_uint64 * 8)()
        _lib.check(_lib.load().fs_index_share_counts(self._h, out), "fs_index_share_counts")
        names = ("windows", "windows_with_a_key_in_the_filter", "map_entries", "pairs_tested", "distances",
                 "windows_flagged", "windows_flagged_as_they_are")
        return {k: int(out[i]) for i, k in enumerate(names)}

    def profile(self, corpus, rows_ptr, cap):
        """Diagnostics: one search of `corpus` (records to the device buffer at `rows_ptr`)
        with a HIP event behind each of its kernels; returns [(kernel name, ms), ...] in
        launch order.  The GPU should be idle otherwise (fs_search_profile)."""
        names = C.create_string_buffer(2048)
        ms = (C.c_double * 32)()
        n = C.c_uint32()
        _lib.check(_lib.load().fs_search_profile(self._h, corpus._h, C.c_void_p(rows_ptr), cap, 1,
                                                 names, len(names), ms, 32, C.byref(n)),
                   "fs_search_profile")
        labels = names.value.decode().split("\n")
        return [(labels[i], float(ms[i])) for i in range(min(n.value, 32, len(labels) - 1))]

    def reload_switches(self):
        """Diagnostics: re-read the FS_* environment switches (read at creation)."""
        _lib.check(_lib.load().fs_index_reload_switches(self._h), "fs_index_reload_switches")

    def search_begin(self, corpus, rows_ptr, cap, packed=False, header=False):
        """Queue a search (rows to the device buffer at `rows_ptr`) and return a
        ticket for search_end; up to four may be in flight per index.  `header`:
        the buffer starts with a 32-byte header whose first eight bytes receive the
        record count, the `cap` records follow it."""
        t = C.c_uint32(0)
        mode = self._rows_mode(packed) | (abi.FS_ROWS_HEADER if header else 0)
        _lib.check(_lib.load().fs_search_corpus_begin(
            self._h, corpus._h, C.c_void_p(rows_ptr), int(cap), mode, C.byref(t)),
            "fs_search_corpus_begin")
        return t.value

    def search_end(self, ticket):
        """(n_rows, stats) of a queued search; FsError(FS_E_CAPACITY).required when
        the row buffer was too small."""
        st = abi.FsStats()
        n = C.c_uint64(0)
        rc = _lib.load().fs_search_corpus_end(self._h, int(ticket), C.byref(n), C.byref(st))
        if rc == abi.FS_E_CAPACITY:
            err = _lib.FsError(rc, "fs_search_corpus_end", "row buffer too small")
            err.required = int(n.value)
            raise err
        _lib.check(rc, "fs_search_corpus_end")
        return int(n.value), st

    def unpack_device(self, packed_ptr, n, rows_ptr):
        """Expand `n` 16-byte wire records at device address `packed_ptr` into
        fs_row records at `rows_ptr` (both on this index's device)."""
        _lib.check(_lib.load().fs_rows_unpack(self._h, C.c_void_p(packed_ptr), int(n),
                                              C.c_void_p(rows_ptr)), "fs_rows_unpack")

    def unpack8_device(self, packed_ptr, n, work_off_ptr, n_works, rows_ptr):
        """Expand `n` 8-byte wire records; `work_off_ptr`: device address of the
        n_works + 1 uint64 work offsets of the batch the records come from."""
        _lib.check(_lib.load().fs_rows_unpack8(self._h, C.c_void_p(packed_ptr), int(n),
                                               C.c_void_p(work_off_ptr), int(n_works),
                                               C.c_void_p(rows_ptr)), "fs_rows_unpack8")

    def reuse_histogram_device(self, rows_ptr, n_rows, thresholds):
        """`format` aggregation over device-resident fs_row records (after a
        search or a gather): counts[n_script][len(thresholds) + 1] on the host."""
        import torch
        thr = np.ascontiguousarray(thresholds, dtype=np.float64)
        n_script = int(self.info["n_script"])
        out = torch.zeros(max(1, n_script) * (len(thr) + 1), dtype=torch.int32, device="cuda")
        torch_ready()                       # (the zeros are there before the kernel adds to them)
        _lib.check(_lib.load().fs_reuse_histogram_rows(
            self._h, C.c_void_p(rows_ptr), int(n_rows), abi.ptr(thr, C.c_double), len(thr),
            C.c_void_p(out.data_ptr())), "fs_reuse_histogram_rows")
        return out.cpu().numpy().view(np.uint32)[:n_script * (len(thr) + 1)] \
            .reshape(n_script, len(thr) + 1)

    def scan_benchmark(self, corpus, reps=20):
        """Average milliseconds of one scan-kernel launch over `corpus`."""
        ms = C.c_double(0)
        _lib.check(_lib.load().fs_scan_benchmark(self._h, corpus._h, reps, C.byref(ms)),
                   "fs_scan_benchmark")
        return ms.value

    def close(self):
        """Destroys the index; corpora created on it are closed first (the library
        would only detach them)."""
        for c in list(getattr(self, "_corpora", ())):
            c.close()
        if self._h:
            _lib.load().fs_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
