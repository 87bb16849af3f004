"""Native text front end: fan works -> string ids on host threads (fs_textenc_*, csrc/fs_text.hip).

The reference reads and tokenises every work in Python inside its worker pool
(/root/reference/search.py:164-166).  spaCy's tokenizer -- and tokenizer.py, its restatement
here -- splits a text on whitespace, treats every chunk by itself and caches chunk -> tokens.
The native encoder is that cache and the splitting: it reads the files, splits them and
looks every chunk up, on all cores; a chunk it has not been taught comes back as a
placeholder, `tokenizer.tokenize` (the rules, and the oracle of the native path in
tests/test_textenc.py) is run on it once, and the encoder learns the answer.  The token stream
is therefore the rule tokenizer's by construction.  Works of 100000 bytes or more (the
reference cuts those into pieces first, search.py:47-63, vocab.chunk_text) and files that are
not valid UTF-8 are left to the Python path, work by work.

FANDOM_SEARCH_NATIVE_TEXT=0 switches it off (search.py then tokenises in its process pool
as before); it is used with the default rule tokenizer only (FANDOM_SEARCH_TOKENIZER unset or
"rules")."""

import ctypes as C
import os
import threading

import numpy as np

from . import _lib

PLACEHOLDER = 0x80000000


def enabled():
    return (os.environ.get("FANDOM_SEARCH_NATIVE_TEXT", "1") != "0"
            and os.environ.get("FANDOM_SEARCH_TOKENIZER", "rules") == "rules")


def _bind(L):
    if getattr(L, "_textenc_bound", False):
        return L
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.fs_textenc_create.restype = C.c_int
    L.fs_textenc_create.argtypes = [C.POINTER(C.c_void_p)]
    L.fs_textenc_destroy.restype = None
    L.fs_textenc_destroy.argtypes = [C.c_void_p]
    L.fs_textenc_add.restype = C.c_int
    L.fs_textenc_add.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_uint64, u64p, u32p]
    L.fs_textenc_encode_files.restype = C.c_int
    L.fs_textenc_encode_files.argtypes = [
        C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(u32p), u64p, C.POINTER(u64p),
        C.POINTER(C.POINTER(C.c_int32)), C.POINTER(u8p), C.POINTER(u64p), u64p]
    L.fs_textenc_encode_files_vec.restype = C.c_int
    L.fs_textenc_encode_files_vec.argtypes = [
        C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, u32p, C.c_uint64, C.POINTER(u32p), u64p, C.POINTER(u64p),
        C.POINTER(C.POINTER(C.c_int32)), C.POINTER(u8p), C.POINTER(u64p), u64p,
        C.POINTER(u32p), u64p, C.POINTER(C.c_int32)]
    L._textenc_bound = True
    return L


class TextEncoder(object):
    """encode_files(filenames) -> (token counts per file, string ids of all tokens) against
    `vocab` (which grows by the strings it had not seen).  start(filenames) runs the native part
    of a later encode_files on a background thread (the GIL is released inside the library)."""

    def __init__(self, vocab, threads=0):
        self.vocab = vocab
        if not threads:
            from .search import usable_cpus
            threads = min(16, usable_cpus())
        self.threads = int(threads)
        self._L = _bind(_lib.load())
        # native encoders, each with a table of its own (taught alike) and a lock: start() gives
        # consecutive batches to them in turn, so that one batch's serial parts (its threads'
        # start, the copies of its results) run beside the other's parallel part
        # (FANDOM_SEARCH_TEXT_HANDLES, default 2)
        self._hs = []
        for _ in range(max(1, int(os.environ.get("FANDOM_SEARCH_TEXT_HANDLES", "2")))):
            h = C.c_void_p()
            _lib.check(self._L.fs_textenc_create(C.byref(h)), "fs_textenc_create")
            self._hs.append((h, threading.Lock()))
        self._turn = 0
        self._pending = {}
        self._taught = 0
        self.last_vec = None
        self._teach_plain()

    def close(self):
        hs, self._hs = getattr(self, "_hs", []), []
        for h, _ in hs:
            self._L.fs_textenc_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- what the encoder knows -------------------------------------------------------------

    def _add(self, chunks, pieces):
        """Teach: chunks[i] (str) tokenises into the string ids pieces[i]."""
        if not chunks:
            return
        raw = [c.encode("utf-8") for c in chunks]
        coff = np.zeros(len(raw) + 1, dtype=np.uint64)
        coff[1:] = np.cumsum([len(b) for b in raw], dtype=np.uint64)
        poff = np.zeros(len(raw) + 1, dtype=np.uint64)
        poff[1:] = np.cumsum([len(p) for p in pieces], dtype=np.uint64)
        flat = np.fromiter((s for p in pieces for s in p), dtype=np.uint32, count=int(poff[-1]))
        blob = b"".join(raw)
        for h, lock in self._hs:
            with lock:
                _lib.check(self._L.fs_textenc_add(
                    h, blob, coff.ctypes.data_as(C.POINTER(C.c_uint64)), len(raw),
                    poff.ctypes.data_as(C.POINTER(C.c_uint64)), flat.ctypes.data_as(C.POINTER(C.c_uint32))),
                    "fs_textenc_add")

    def _teach_plain(self):
        """The vocabulary's words that are their own single token (letters only, no special
        case: tokenizer.tokenize's own shortcut), from where the last call stopped."""
        from . import tokenizer
        strings = self.vocab.strings
        new = [(sid, w) for sid, w in enumerate(strings[self._taught:], self._taught)
               if w.isalpha() and w not in tokenizer.SPECIAL_CASES]
        self._taught = len(strings)
        self._add([w for _, w in new], [(sid,) for sid, _ in new])

    # ---- encoding ---------------------------------------------------------------------------

    def _native(self, filenames, made=None, turn=0):
        """(tokens, work offsets, status per file, unknown chunks' texts).  `made`: a dict that
        receives "vec" = (vector id per token, OOV tokens, string ids == vector ids) when the
        batch needs nothing more from the host -- no unknown chunk, no work left to the Python
        path --, made by the encoding threads themselves (fs_textenc_encode_files_vec)."""
        paths = b"".join(os.fsencode(f) + b"\0" for f in filenames)
        tok, woff = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)()
        status, ub, uo = C.POINTER(C.c_int32)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)()
        n_tok, n_unk = C.c_uint64(), C.c_uint64()
        vec, n_oov, same = C.POINTER(C.c_uint32)(), C.c_uint64(), C.c_int32()
        h, lock = self._hs[turn % len(self._hs)]
        with lock:
            if made is not None:
                vt = self.vocab.vec_ids()          # (covers every string id the encoder has been taught)
                _lib.check(self._L.fs_textenc_encode_files_vec(
                    h, paths, len(filenames), self.threads, vt.ctypes.data_as(C.POINTER(C.c_uint32)), len(vt),
                    C.byref(tok), C.byref(n_tok), C.byref(woff), C.byref(status), C.byref(ub), C.byref(uo),
                    C.byref(n_unk), C.byref(vec), C.byref(n_oov), C.byref(same)), "fs_textenc_encode_files_vec")
            else:
                _lib.check(self._L.fs_textenc_encode_files(
                    h, paths, len(filenames), self.threads, C.byref(tok), C.byref(n_tok), C.byref(woff),
                    C.byref(status), C.byref(ub), C.byref(uo), C.byref(n_unk)), "fs_textenc_encode_files")
            n = len(filenames)

            def arr(ptr, count, dtype):          # (an empty vector's data() may be NULL)
                return np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count and ptr else np.zeros(count, dtype)

            t = arr(tok, n_tok.value, np.uint32)
            w = arr(woff, n + 1, np.uint64)
            st = arr(status, n, np.int32)
            if made is not None and n_unk.value == 0 and n_tok.value and not st.any():
                made["vec"] = (arr(vec, n_tok.value, np.uint32), int(n_oov.value), bool(same.value))
            uoff = arr(uo, n_unk.value + 1, np.uint64)
            ubytes = bytes(arr(ub, int(uoff[-1]) if len(uoff) else 0, np.uint8))
        unk = [ubytes[int(uoff[i]):int(uoff[i + 1])].decode("utf-8") for i in range(n_unk.value)]
        return t, w, st, unk

    def start(self, filenames):
        key = tuple(filenames)
        if not key or key in self._pending:
            return
        box = {}
        turn = self._turn
        self._turn += 1

        def run():
            try:
                # (a batch the encoder knows every chunk of -- any batch, once the first few have
                # taught it -- comes with its vector ids, made by the encoding threads: the search
                # then has no pass of its own to make over the batch's tokens)
                box["r"] = self._native(key, box, turn)
            except BaseException as e:          # (raised again in the caller's thread)
                box["e"] = e

        th = threading.Thread(target=run, daemon=True)
        th.start()
        self._pending[key] = (th, box)
        while len(self._pending) > 3:            # lists nobody asked for
            self._pending.pop(next(iter(self._pending)))

    def encode_files(self, filenames):
        from . import tokenizer
        job = self._pending.pop(tuple(filenames), None)
        self.last_vec = None                     # (tokens, vector ids, OOV tokens, ids equal) of this call, where the encoder made them
        if job is not None:
            job[0].join()
            if "e" in job[1]:
                raise job[1]["e"]
            tok, woff, status, unk = job[1]["r"]
            if "vec" in job[1]:                  # (nothing unknown, nothing left to the Python path: `tok` goes out as it is)
                self.last_vec = (tok,) + job[1]["vec"]
                return np.diff(woff).astype(np.int64), tok
        else:
            box = {}
            tok, woff, status, unk = self._native(filenames, box)
            if "vec" in box:
                self.last_vec = (tok,) + box["vec"]
                return np.diff(woff).astype(np.int64), tok
        v = self.vocab
        for i in np.nonzero(status < 0)[0]:
            raise OSError(-int(status[i]), os.strerror(-int(status[i])), filenames[int(i)])
        lens = np.diff(woff).astype(np.int64)
        if unk:
            # the rules, once per distinct chunk; the encoder learns the answers
            texts = list(dict.fromkeys(unk))
            pieces = {}
            for c in texts:
                pieces[c] = tuple(v.string_id(t) for t in tokenizer.tokenize(c))
            keep = [c for c in texts if pieces[c]]
            self._add(keep, [pieces[c] for c in keep])
            plen = np.fromiter((len(pieces[c]) for c in unk), dtype=np.int64, count=len(unk))
            poff = np.zeros(len(unk) + 1, dtype=np.int64)
            poff[1:] = np.cumsum(plen)
            flat = np.fromiter((s for c in unk for s in pieces[c]), dtype=np.uint32, count=int(poff[-1]))
            ph = np.nonzero(tok >= PLACEHOLDER)[0]
            which = (tok[ph] & np.uint32(PLACEHOLDER - 1)).astype(np.int64)
            counts = np.ones(len(tok), dtype=np.int64)
            counts[ph] = plen[which]
            ends = np.cumsum(counts)
            out = np.repeat(tok, counts)
            reps = plen[which]
            dst = np.repeat(ends[ph] - reps, reps) + (np.arange(int(reps.sum())) - np.repeat(np.cumsum(reps) - reps, reps))
            src = np.repeat(poff[which], reps) + (np.arange(int(reps.sum())) - np.repeat(np.cumsum(reps) - reps, reps))
            out[dst] = flat[src]
            bounds = np.concatenate([[0], ends])[woff.astype(np.int64)]
            tok, lens = out, np.diff(bounds)
        if len(v.strings) != self._taught:
            self._teach_plain()
        left = np.nonzero(status == 1)[0]
        if len(left):
            # long works and undecodable files: the Python path, work by work
            from .search import read_work_tokens
            starts = np.concatenate([[0], np.cumsum(lens)])
            parts, at = [], 0
            for i in left:
                i = int(i)
                ids = np.fromiter((v.string_id(t) for t in read_work_tokens(filenames[i])), dtype=np.uint32)
                parts.append(tok[at:int(starts[i])])
                parts.append(ids)
                at = int(starts[i + 1])
                lens[i] = len(ids)
            parts.append(tok[at:])
            tok = np.concatenate(parts) if parts else tok
            if len(v.strings) != self._taught:
                self._teach_plain()
        return lens, np.ascontiguousarray(tok, dtype=np.uint32)
