"""TEST INFRASTRUCTURE ONLY -- literal CPU restatement of the reference's
6-gram search path (/root/reference/search.py:65-226, 331-334).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the
product path (fandom_search_amd) never does.

PARITY UNPINNED: the reference cannot be imported here (nearpy, spacy,
en_core_web_md, Levenshtein are not installed) and it
ships no tests, fixtures or golden vectors, so this restatement is pinned by
hand-derived known answers only (tests/test_oracle_known_answers.py).

The reference is non-deterministic as shipped (unseeded LSH hyperplanes,
search.py:114-115; salted hash() for out-of-vocabulary words, search.py:79-83;
os.listdir order, search.py:349).  Here those three are explicit inputs.

Tokens are duck-typed like spaCy's: .vector (float32[D]), .has_vector,
.is_space, .orth_, .orth, .lower_, .lower and str(tok) == tok.orth_.
"""

import csv
from collections import defaultdict
from operator import itemgetter

import numpy

from . import nearpy_restated as nearpy

FIELDS = ['FAN_WORK_FILENAME',           # search.py:20-33
          'FAN_WORK_WORD_INDEX',
          'FAN_WORK_WORD',
          'FAN_WORK_ORTH_ID',
          'ORIGINAL_SCRIPT_WORD_INDEX',
          'ORIGINAL_SCRIPT_WORD',
          'ORIGINAL_SCRIPT_ORTH_ID',
          'ORIGINAL_SCRIPT_CHARACTER',
          'ORIGINAL_SCRIPT_SCENE',
          'BEST_MATCH_DISTANCE',
          'BEST_LEVENSHTEIN_DISTANCE',
          'BEST_COMBINED_DISTANCE']


class Tok(object):
    """Minimal stand-in for a spaCy Token."""
    __slots__ = ("orth_", "orth", "lower_", "lower", "vector", "has_vector",
                 "is_space")

    def __init__(self, orth_, orth, lower_, lower, vector, has_vector=True,
                 is_space=False):
        self.orth_ = orth_
        self.orth = orth
        self.lower_ = lower_
        self.lower = lower
        self.vector = vector
        self.has_vector = has_vector
        self.is_space = is_space

    def __str__(self):
        return self.orth_

    __repr__ = __str__   # spaCy: Token.__repr__ returns the text in Python 3


def lev_distance(a, b):
    """Levenshtein.distance (python-Levenshtein, search.py:14,190): unit-cost
    insert / delete / substitute over Unicode code points."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1,
                           prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def mk_vectors(sp_txt, oov_hash):
    """search.py:65-84.  `oov_hash` replaces Python's salted hash()."""
    rows = len(sp_txt)
    cols = len(sp_txt[0].vector) if rows else 0

    vectors = numpy.empty((rows, cols), dtype=float)
    for i, word in enumerate(sp_txt):
        if word.has_vector:
            vectors[i] = word.vector
        else:
            w_str = str(word)
            vectors[i] = 0
            vectors[i][oov_hash(w_str) % cols] = 1.0
            vectors[i][oov_hash(w_str * 2) % cols] = 1.0
            vectors[i][oov_hash(w_str * 3) % cols] = 1.0
    return vectors


def _windows(vectors, window_size):
    """search.py:94-95 / 170-173, kept as (n, D) views instead of ravel()ed
    copies; LiteralArith ravels them, which gives the reference's layout."""
    return [vectors[i:i + window_size, :]
            for i in range(vectors.shape[0] - window_size + 1)]


def build_lsh_engine(orig, window_size, number_of_hashes, hash_dimensions,
                     normals, oov_hash, arith, unique_filter=False):
    """search.py:86-124.  `orig` is the list of lower-cased script tokens
    (the reference's Doc(vocab, word_lowercase), search.py:151);
    normals[h] is the (hash_dimensions, D*n) matrix of hash h."""
    orig_vectors = mk_vectors(orig, oov_hash)
    orig_win_vectors = _windows(orig_vectors, window_size)

    hashes = []
    for i in range(number_of_hashes):
        h = nearpy.RandomBinaryProjections('rbp{}'.format(i), hash_dimensions,
                                           normals[i], arith)
        hashes.append(h)

    engine = nearpy.Engine(hashes, arith, unique_filter=unique_filter)

    for ix, row in enumerate(orig_win_vectors):
        # str(Doc[ix:ix+n]) == tokens joined by single spaces
        span = ' '.join(str(t) for t in orig[ix: ix + window_size])
        engine.store_vector(row, (ix, span))
    return engine


class AnnIndexSearch(object):
    """search.py:130-226 with injected inputs.

    script_rows: list of [LOWERCASE, SPACY_ORTH_ID, SCENE, CHARACTER]
                 (load_markup_script output without its header, search.py:302)
    script_toks: lower-cased script tokens (vectors for the index)
    """

    def __init__(self, script_rows, script_toks, window_size,
                 number_of_hashes, hash_dimensions, distance_threshold,
                 normals, oov_hash=None, arith=None, unique_filter=False):
        orig_csv = [[i] + list(r) for i, r in enumerate(script_rows)]
        (self.word_index,
         self.word_lowercase,
         self.orth_id,
         self.scene,
         self.character) = zip(*orig_csv)

        self.window_size = window_size
        self.distance_threshold = distance_threshold
        self.oov_hash = oov_hash or (lambda s: 0)
        self.arith = arith or nearpy.LiteralArith()
        self.engine = build_lsh_engine(script_toks, window_size,
                                       number_of_hashes, hash_dimensions,
                                       normals, self.oov_hash, self.arith,
                                       unique_filter)
        self.reset_stats()

    def reset_stats(self):
        self._windows_processed = 0

    @property
    def windows_processed(self):
        return self._windows_processed

    def search(self, filename, fan):
        """One fan work (search.py:163-226).  `fan` is the work's token list
        with is_space tokens already dropped (search.py:164-166).

        Steps, in the reference's order:
          :169-173  token vectors -> sliding windows
          :176-184  per window: engine.neighbours, keep distance < threshold
          :188-190  one Levenshtein per kept match, script span text against
                    str(list of fan tokens) == '[T1, T2, ...]'
          :192-218  n word-level records per match, keyed (filename, word ix)
          :224-226  per key the first record of minimal combined distance;
                    records sorted
        """
        n = self.window_size
        wins = _windows(mk_vectors(fan, self.oov_hash), n)

        per_word = defaultdict(list)
        for fan_ix, win in enumerate(wins):
            self._windows_processed += 1
            kept = [(data[0], data[1], dist)
                    for _vec, data, dist in self.engine.neighbours(win)
                    if dist < self.distance_threshold]
            for match_ix, match_str, dist in kept:
                fan_context = str(fan[fan_ix:fan_ix + n])
                lev_d = lev_distance(match_str, fan_context)
                for k in range(n):
                    f = fan_ix + k
                    o = match_ix + k
                    per_word[(filename, f)].append(
                        [filename, f, fan[f].orth_, fan[f].orth,
                         o, self.word_lowercase[o], self.orth_id[o],
                         self.character[o], self.scene[o],
                         dist, lev_d, dist * lev_d])

        best = [min(cands, key=itemgetter(11)) for cands in per_word.values()]
        return sorted(best)


def write_records(records, filename):
    """search.py:331-334."""
    with open(filename, 'w', encoding='utf-8') as out:
        wr = csv.writer(out)
        wr.writerows(records)
