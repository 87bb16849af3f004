"""Hand-derived known answers that pin the oracle (the reference ships no
tests or fixtures): every expectation below follows from
reading /root/reference/search.py and the NearPy / Levenshtein semantics of
SURVEY.md 2.3, not from running the oracle."""

import random

import numpy as np
import pytest

from fandom_search_amd import abi, synth, vocab
from oracle import nearpy_restated as nr
from oracle import search_restated as sr
from tests import util


def test_seeded_shuffle_matches_cpython():
    """search.py:354-355 on CPython 3.10 (measured in SURVEY.md section 4)."""
    x = list(range(10))
    random.seed(4815162342)
    random.shuffle(x)
    assert x == [5, 4, 2, 3, 1, 8, 7, 0, 6, 9]


def test_levenshtein_of_verbatim_ngram_is_n_plus_1():
    """search.py:189-190: str(list) adds 2 brackets and n-1 commas."""
    for n in (2, 4, 6, 10):
        ws = ["w%d" % i for i in range(n)]
        assert sr.lev_distance(" ".join(ws), "[" + ", ".join(ws) + "]") == n + 1
    assert sr.lev_distance("kitten", "sitting") == 3
    assert sr.lev_distance("", "abc") == 3
    assert sr.lev_distance("flaw", "lawn") == 2
    # the two examples in python-Levenshtein's own docstring of distance()
    assert sr.lev_distance("Hello world!", "Holly grail!") == 7
    assert sr.lev_distance("Brian", "Jesus") == 5
    # textbook pairs: insertion/deletion/substitution mixes, code points not bytes
    assert sr.lev_distance("Saturday", "Sunday") == 3
    assert sr.lev_distance("intention", "execution") == 5
    assert sr.lev_distance("gumbo", "gambol") == 2
    assert sr.lev_distance("\u00e9t\u00e9", "ete") == 2


def test_spacy_string_hash_known_values():
    """Values printed in spaCy's documentation ("Vocab, hashes and lexemes": the rows of
    "I love coffee"; StringStore: nlp.vocab.strings["apple"])."""
    assert vocab.hash_string("coffee") == 3197928453018144401
    assert vocab.hash_string("apple") == 8566208034543834098
    assert vocab.hash_string("I") == 4690420944186131903
    assert vocab.hash_string("love") == 3702023516439754181


def test_unique_filter_keeps_first_insertion_order():
    f = nr.UniqueFilter()
    items = [("v", (3, "a")), ("v", (1, "b")), ("v", (3, "a")), ("v", (2, "c")), ("v", (1, "b"))]
    assert [d for _, d in f.filter_vectors(items)] == [(3, "a"), (1, "b"), (2, "c")]


def test_nearest_filter_is_stable_and_keeps_n():
    f = nr.NearestFilter(3)
    items = [("v", i, d) for i, d in enumerate([0.5, 0.1, 0.1, 0.7, 0.1, 0.0])]
    assert [i for _, i, _ in f.filter_vectors(items)] == [5, 1, 2]


def test_binary_projection_key_is_strict_sign():
    normals = np.array([[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [1.0, 1.0]])
    h = nr.RandomBinaryProjections("h", 4, normals, nr.LiteralArith())
    assert h.hash_vector(np.array([[2.0, 0.0]])) == ["1001"]      # 0.0 is not > 0.0
    assert h.hash_vector(np.array([[-1.0, 3.0]])) == ["0111"]


def test_oov_vector_is_three_hot():
    """search.py:79-83 with an injected hash."""
    table = {"zz": 5, "zzzz": 299 + 300, "zzzzzz": 5}
    t = sr.Tok("zz", 1, "zz", 1, np.zeros(300, np.float32), has_vector=False)
    v = sr.mk_vectors([t], lambda s: table[s])
    assert v.shape == (1, 300) and v.dtype == np.float64
    assert v[0].sum() == 2.0 and v[0][5] == 1.0 and v[0][299] == 1.0   # two hashes coincide


def _one_plant_case():
    words = synth.vocab_words()
    emb = synth.embedding()
    voc = vocab.Vocab(words, emb)
    rng = np.random.default_rng(3)
    script = rng.permutation(4000)[:300].astype(np.uint32)           # all distinct tokens
    fan = rng.permutation(np.arange(4000, 8000))[:120].astype(np.uint32)
    fan[40:48] = script[100:108]                                     # 8 verbatim tokens
    return words, emb, voc, script, fan


@pytest.mark.parametrize("arith", [nr.LiteralArith, nr.CanonicalArith])
def test_single_verbatim_span(arith):
    """8 verbatim tokens = 3 matching windows covering 8 fan words: one row per
    covered word, script index = fan index + 60, Levenshtein 7, distance ~ 0."""
    words, emb, voc, script, fan = _one_plant_case()
    scene, char = synth.script_columns(len(script))

    def toks(ids):
        return [sr.Tok(words[i], voc.orth(i), words[i], voc.orth(i), emb[i]) for i in ids]

    rows = [[words[t], voc.orth(int(t)), int(scene[i]), char[i]] for i, t in enumerate(script)]
    idx = sr.AnnIndexSearch(rows, toks(script), 6, 15, 14, 0.1, synth.lsh_normals(6), arith=arith())
    out = idx.search("f.txt", toks(fan))
    assert idx.windows_processed == 120 - 6 + 1
    assert [r[1] for r in out] == list(range(40, 48))
    for r in out:
        assert r[0] == "f.txt" and r[4] == r[1] + 60
        assert r[2] == r[5] == words[fan[r[1]]] and r[3] == r[6] == voc.orth(int(fan[r[1]]))
        assert r[7] == char[r[4]] and r[8] == int(scene[r[4]])
        assert r[10] == 7 and abs(r[9]) < 1e-15 and r[11] == r[9] * 7


def test_c_oracle_matches_python_oracle_bit_for_bit(synth_base):
    words, emb, voc, script, fan = _one_plant_case()
    normals = synth.lsh_normals(6)
    scene, char = synth.script_columns(len(script))

    def toks(ids):
        return [sr.Tok(words[i], voc.orth(i), words[i], voc.orth(i), emb[i]) for i in ids]

    rows = [[words[t], voc.orth(int(t)), int(scene[i]), char[i]] for i, t in enumerate(script)]
    idx = sr.AnnIndexSearch(rows, toks(script), 6, 15, 14, 0.1, normals, arith=nr.CanonicalArith())
    works = [fan, synth.fanwork_tokens(0, 150, script), fan[30:60]]
    py = []
    for w, ids in enumerate(works):
        py += idx.search(w, toks(ids))
    off = np.zeros(len(works) + 1, np.uint64)
    off[1:] = np.cumsum([len(w) for w in works])
    cfg = abi.make_config()
    oi = util.oracle_index(cfg, script, words, emb, normals, threads=2)
    got, st = oi.search(np.concatenate(works), off, synth_base["chars"], synth_base["off"])
    assert len(got) == len(py) > 0
    for a, b in zip(py, got):
        assert (a[0], a[1], a[4], a[10]) == (b["work"], b["fan_ix"], b["orig_ix"], b["lev"])
        assert a[9] == b["dist"] and a[11] == b["comb"]
    assert st.candidates == idx.engine.candidate_count
    for w in (0, 17, 294):
        win = np.asarray([emb[t] for t in script[w:w + 6]], dtype=float)
        assert list(oi.script_keys(w)) == [int(h.hash_vector(win)[0], 2) for h in idx.engine.lshashes]


def test_near_miss_does_not_match():
    """One substituted token inside a 6-gram: cos <= (5 + c_max)/6 < 0.9 for the
    synthetic table (c_max = 0.314), so no window covering it may match."""
    words, emb, voc, script, fan = _one_plant_case()
    fan = fan.copy()
    fan[40:48] = script[100:108]
    fan[43] = 7999                                   # breaks every window over word 43
    cfg = abi.make_config()
    oi = util.oracle_index(cfg, script, words, emb, synth.lsh_normals(6), threads=1)
    chars, coff = vocab.pack_strings(words)
    got, _ = oi.search(fan, np.array([0, len(fan)], np.uint64), chars, coff)
    assert len(got) == 0
