"""Vocabulary, string ids and tokenizer for the search path.

The reference gets all of this from spaCy + en_core_web_md
(/root/reference/search.py:40-63, 166, 322-327), neither of which is available
here, so the host layer carries its own:

* string ids    spaCy's `Token.orth` / `.lower` are MurmurHash64A(utf8, seed=1)
                of the text; `hash_string` reproduces that (known answers:
                'coffee' -> 3197928453018144401, 'apple' -> 8566208034543834098)
                so the two *_ORTH_ID columns of the match CSV carry the values
                a spaCy run would write.
* vectors       a word -> row table into an (V, D) float32 matrix (spaCy's
                key2row + vectors.data); words without a row take the
                reference's out-of-vocabulary construction (search.py:79-83):
                zeros with 1.0 at hash(w) % D, hash(w*2) % D, hash(w*3) % D.
                Python's hash() is salted per process, so the hash is an
                explicit, seeded input here (`oov_hash`).
* vector ids    what the device scans.  Row r of the matrix is id r; an OOV
                vector is id OOV_FLAG | ((a*D + b)*D + c) with a <= b <= c the
                sorted hot positions, so two tokens carry the same id exactly
                when they carry the same vector.
* tokenizer     fandom_search_amd.tokenizer: spaCy 2.x's tokenizer algorithm with
                its English prefix / suffix / infix rules and tokenizer
                exceptions restated (contractions, abbreviations, URLs ...).
                FANDOM_SEARCH_TOKENIZER=spacy uses an installed spaCy instead,
                =simple the old `\\w+|[^\\w\\s]` splitter.  On the synthetic corpora
                (single-space separated alphabetic words) all three agree.
"""

import re

import numpy as np

_M = 0xC6A4A7935BD1E995
_MASK = (1 << 64) - 1
OOV_FLAG = 0x80000000


def murmurhash64a(data, seed=1):
    n = len(data)
    h = (seed ^ (n * _M)) & _MASK
    body = n - n % 8
    for i in range(0, body, 8):
        k = int.from_bytes(data[i:i + 8], "little")
        k = (k * _M) & _MASK
        k ^= k >> 47
        k = (k * _M) & _MASK
        h ^= k
        h = (h * _M) & _MASK
    if n % 8:
        h ^= int.from_bytes(data[body:], "little")
        h = (h * _M) & _MASK
    h ^= h >> 47
    h = (h * _M) & _MASK
    h ^= h >> 47
    return h


_HASH_CACHE = {}


def hash_string(text):
    """spaCy StringStore key of `text` (remembered: the CSV writer asks once per record,
    a corpus has far fewer distinct words than records)."""
    h = _HASH_CACHE.get(text)
    if h is None:
        h = murmurhash64a(text.encode("utf8"), 1)
        if len(_HASH_CACHE) < (1 << 20):
            _HASH_CACHE[text] = h
    return h


def default_oov_hash(text):
    """Seeded stand-in for the salted hash() of search.py:81-83."""
    return murmurhash64a(text.encode("utf8"), 0x5EED)


_TOKEN_RE = re.compile(r"\w+|[^\w\s]", re.UNICODE)
_SPACY_NLP = None


def tokenize(text):
    """Token texts of `text`, whitespace dropped (search.py:166 drops
    is_space tokens).  Default: the restated spaCy rules (tokenizer.py)."""
    import os
    kind = os.environ.get("FANDOM_SEARCH_TOKENIZER", "rules")
    if kind == "simple":
        return _TOKEN_RE.findall(text)
    if kind == "spacy":
        global _SPACY_NLP
        if _SPACY_NLP is None:
            import spacy                       # fails loudly when it is not installed
            _SPACY_NLP = spacy.blank("en")
        return [t.text for t in _SPACY_NLP(text) if not t.is_space]
    from . import tokenizer
    return tokenizer.tokenize(text)


def chunk_text(txt, size=100000):
    """Chunking of sp_parse_chunks (search.py:47-63): pieces of at most
    100000 characters cut at a space; the `size` argument is ignored by the
    reference too (100000 is hard-coded at :51,:56).  Held to the reference's own
    function by tests/golden/make_search_golden.py, its two oddities included: a rest
    of exactly 100000 characters indexes one past the end (IndexError, as the reference),
    and where no space lies within 100000 characters the reference's index walks back
    below the chunk's start -- past 0 it wraps round to the end of the text -- and it
    either raises IndexError (no space anywhere) or yields chunks for ever; the endless
    case is an error here."""
    if len(txt) < 100000:
        yield txt
        return
    start = 0
    while start < len(txt):
        end = start + 100000
        if end > len(txt):
            end = len(txt)
        else:
            while txt[end] != ' ':
                end -= 1
                if end < start:
                    if ' ' not in txt:
                        raise IndexError('string index out of range')
                    raise ValueError('no space within 100000 characters at offset %d: the '
                                     'reference (search.py:57-58) does not terminate on this text'
                                     % start)
            if end < start:                    # (a space at start - 1 ... cannot happen: start = end + 1)
                raise ValueError('no space within 100000 characters at offset %d' % start)
        yield txt[start:end]
        start = end + 1


class Vocab(object):
    """Growable string table + fixed vector table."""

    def __init__(self, words, vectors, oov_hash=default_oov_hash, rows=None):
        """`rows` (optional): row of `vectors` per word, as spaCy's key2row -- several words
        may share a row (en_core_web_md keeps 20k rows for 685k keys) and then share a vector
        id; without it word i has row i."""
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        if vectors.ndim != 2:
            raise ValueError("vectors must be (rows, dim)")
        if rows is None:
            if len(words) != vectors.shape[0]:
                raise ValueError("one vector row per word expected (or pass rows=)")
            rows = range(len(words))
        else:
            rows = [int(r) for r in rows]
            if len(rows) != len(words) or (rows and not (0 <= min(rows) and max(rows) < vectors.shape[0])):
                raise ValueError("rows must name a row of the vector table for every word")
        self.vectors = vectors
        self.dim = vectors.shape[1]
        self.key2row = {w: r for w, r in zip(words, rows)}
        self.oov_hash = oov_hash
        self.strings = []
        self._string_id = {}
        self._vec_id = []        # per string id
        self._orth = []          # per string id: spaCy hash
        for w in words:          # (string id == row id while every word has a row of its own)
            self.string_id(w)

    def string_id(self, text):
        sid = self._string_id.get(text)
        if sid is None:
            sid = len(self.strings)
            self._string_id[text] = sid
            self.strings.append(text)
            self._vec_id.append(self._vector_id(text))
            self._orth.append(hash_string(text))
        return sid

    def _vector_id(self, text):
        row = self.key2row.get(text)
        if row is not None:
            return row
        d = self.dim
        hot = sorted((self.oov_hash(text) % d,
                      self.oov_hash(text * 2) % d,
                      self.oov_hash(text * 3) % d))
        return OOV_FLAG | ((hot[0] * d + hot[1]) * d + hot[2])

    def vec_id(self, sid):
        return self._vec_id[sid]

    def vec_ids(self):
        """Vector id per string id as one uint32 array (grown with the string table)."""
        cached = getattr(self, "_vec_id_arr", None)
        if cached is None or len(cached) != len(self._vec_id):
            self._vec_id_arr = cached = np.array(self._vec_id, dtype=np.uint32)
        return cached

    def orth(self, sid):
        return self._orth[sid]

    def has_vector(self, sid):
        return not (self._vec_id[sid] & OOV_FLAG)

    def vector(self, sid):
        """float32 vector of string `sid` (table row, or the OOV 3-hot)."""
        vid = self._vec_id[sid]
        if not vid & OOV_FLAG:
            return self.vectors[vid]
        return oov_vector(vid, self.dim)

    def encode(self, texts):
        """(string ids, vector ids) uint32 arrays for a list of token texts.
        The list is factorised first, so the vocabulary is consulted once per
        distinct string, not once per token."""
        if len(texts) == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        import sys
        pd = sys.modules.get("pandas")
        if pd is None and len(texts) > 200000:          # (pandas costs 0.3 s to import)
            try:
                import pandas as pd
            except ImportError:
                pd = None
        if pd is not None:
            codes, uniques = pd.factorize(np.asarray(texts, dtype=object), sort=False)
        else:
            uniques, codes = np.unique(np.asarray(texts, dtype=object), return_inverse=True)
        usid = np.fromiter((self.string_id(t) for t in uniques), dtype=np.uint32,
                           count=len(uniques))
        uvid = np.fromiter((self._vec_id[s] for s in usid), dtype=np.uint32,
                           count=len(uniques))
        return usid[codes], uvid[codes]

    def string_table(self):
        """UTF-32 code points of all strings + offsets (n_strings + 1); remembered until the
        table grows (a batch of 500 works asks once, and most batches add no string)."""
        cached = getattr(self, "_string_table", None)
        if cached is None or cached[0] != len(self.strings):
            chars, off = pack_strings(self.strings)
            self._string_table = cached = (len(self.strings), chars, off)
        return cached[1], cached[2]


def oov_vector(vid, dim):
    code = vid & ~OOV_FLAG
    c = code % dim
    b = (code // dim) % dim
    a = code // (dim * dim)
    v = np.zeros(dim, dtype=np.float32)
    v[a] = v[b] = v[c] = 1.0
    return v


def pack_strings(strings):
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    if strings:
        off[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64)
    chars = np.frombuffer("".join(strings).encode("utf-32-le"),
                          dtype=np.uint32).copy()
    return chars, off
