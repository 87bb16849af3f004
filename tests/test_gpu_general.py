"""GPU parity of the general (LSH) pipeline: window sizes, vector tables and
tokens for which the exact n-gram proof does not hold.  Rows must be
bit-identical to the plain-C oracle, which runs the reference's algorithm with
no shortcut."""

import numpy as np
import pytest

from fandom_search_amd import abi, synth
from fandom_search_amd.vocab import pack_strings
from tests import util

pytestmark = pytest.mark.gpu


def _run(cfg, script, swords, emb, normals, tok, off, chars, coff, tok_str=None, words=None):
    from fandom_search_amd.engine import ScriptIndex
    from oracle import c_oracle
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    got, st = ix.search(ix.corpus(tok, off, chars, coff, tok_str=tok_str))
    sch, so = pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8)
    want, ost = oi.search(tok, off, chars, coff, tok_str=tok_str)
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and st.windows_processed == ost.windows_processed
    return ix, got, st


@pytest.mark.parametrize("unique", [1, 0])
@pytest.mark.parametrize("n", [8, 10])
def test_window_sizes_without_proof(synth_base, n, unique):
    """BASELINE.json configs[3] (n-gram sweep): for n = 8 and 10 one substituted
    token can stay within the threshold ((n-1+c_max)/n > 0.9), so the index must
    take the LSH pipeline.  `unique`: with and without NearPy's UniqueFilter (1.0.0's
    Engine.neighbours() applies fetch filters only when given one)."""
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(3000)
    tok, off = util.ragged_corpus([700] * 12 + [0, n - 1, n, 1500], script)
    cfg = abi.make_config(window_size=n, unique_filter=unique)
    ix, got, st = _run(cfg, script, [words[int(t)] for t in script], emb, synth.lsh_normals(n),
                       tok, off, synth_base["chars"], synth_base["off"])
    assert ix.info["proof_ok"] == 0 and st.path == abi.FS_MODE_GENERAL
    assert len(got) > 0


@pytest.mark.parametrize("unique", [1, 0])
@pytest.mark.parametrize("n", [8, 9, 10, 12])
def test_one_slot_prefilter_of_the_lsh_pipeline(synth_base, monkeypatch, n, unique):
    """Where the proof fails by one slot only, k_scan_near flags the windows that equal a
    script window in all but one slot (failed 3-gram tests confined to three consecutive
    positions) and the LSH work runs on those; FS_LSH_PREFILTER=0 computes keys and
    buckets for every window.  Same bytes, equal to the oracle, with planted spans whose
    odd token sits at every slot of a window (records with a distance well above 0).
    `unique` = 0: without NearPy's UniqueFilter a window comes back once per table whose
    bucket holds it, the ten nearest are then copies of the nearest few (the
    whole shortcut stack -- per-n-gram records, one-slot map, sift -- under both settings)."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(4000)
    tok, off = util.ragged_corpus([800] * 14 + [0, n - 1, n, 1700], script)
    tok = tok.copy()
    # one substituted token at slot j, replaced by the table vector closest to it (cosine
    # 0.25-0.30 on this table: (n-1+c)/n stays above 0.9, a genuine inexact neighbour);
    # a random substitute (cosine ~0) only produces a window the prefilter must let through
    cos = emb[script[:600]] @ emb.T
    cos[np.arange(600), script[:600]] = -1.0
    best = cos.argmax(axis=1)
    for j in range(n):
        src = 300 + 20 * j
        at = int(off[j % 14]) + 100 + 40 * j
        tok[at:at + n] = script[src:src + n]
        tok[at + j] = best[src + j] if j % 2 == 0 else (int(tok[at + j]) + 17) % len(words)
    for j in range(3):                                   # two substituted tokens: no match
        at = int(off[10 + j]) + 600
        tok[at:at + n] = script[900 + 30 * j:900 + 30 * j + n]
        tok[at + 1] = (int(tok[at + 1]) + 5) % len(words)
        tok[at + n - 2] = (int(tok[at + n - 2]) + 9) % len(words)
    cfg = abi.make_config(window_size=n, unique_filter=unique)
    normals = synth.lsh_normals(n)
    # (script words as written in the script, not always the table's spelling: the
    # Levenshtein distance of an identical-id match is not a constant)
    swords = [words[int(t)].upper() if i % 7 == 0 else words[int(t)] for i, t in enumerate(script)]
    ix, got, st = _run(cfg, script, swords, emb, normals, tok, off,
                       synth_base["chars"], synth_base["off"])
    assert st.path == abi.FS_MODE_GENERAL
    assert ix.kernel_name(ix.corpus(tok, off, synth_base["chars"], synth_base["off"])) == "k_near_sift<%d>" % n
    # (what k_lsh_sift leaves to the wave-per-window kernel: the windows one slot away from a
    # script n-gram that may be within the threshold, a small share of the candidates)
    assert 0 < st.lsh_pending < st.candidates
    assert int((got["dist"] > 0.01).sum()) > 0           # inexact neighbours are records
    assert len(set(got["lev"].tolist())) > 1
    # the windows with a script n-gram's ids take the n-gram's record of this string table
    # (k_lsh_gramtab, built with the corpus); a second search gives the same bytes
    c0 = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    first, st_a = ix.search(c0)
    again, st_b = ix.search(c0)
    assert first.tobytes() == got.tobytes() == again.tobytes() and st_a.matches == st_b.matches == st.matches
    c0.close()
    # without the 3-gram prefilter (keys and buckets for every window), without the
    # wildcard-key filter in front of k_lsh_verify, with every Levenshtein distance
    # computed per match, without the per-n-gram records, without the exact one-slot map: the
    # same bytes
    # ... and with the four-tokens-per-lane form of the prefilter scan (its own 3-gram hash)
    # ... with round 4's chain behind the prefilter scan (bitmap, k_expand, k_lsh_sift over every
    # candidate) instead of k_near_sift + k_lsh_sift2, the same switches under it, and with
    # k_near_sift's lists starting at two entries per wave range (the search reports what it
    # needs and is repeated)
    fused = "k_near_sift<%d>" % n
    chain = "k_scan_near8<%d>" % n
    for env, kernel in (({"FS_LSH_PREFILTER": "0"}, "k_lsh_scan"), ({"FS_LSH_WILD": "0"}, fused),
                        ({"FS_LSH_SELFLEV": "0"}, fused), ({"FS_LSH_GRAMTAB": "0"}, fused),
                        ({"FS_LSH_WMAP": "0"}, fused), ({"FS_SCAN_NEAR8": "0"}, "k_scan_near<%d>" % n),
                        ({"FS_LSH_LEV_LANE": "0"}, fused), ({"FS_NEAR_FUSED": "0"}, chain),
                        ({"FS_NEAR_FUSED": "0", "FS_LSH_WILD": "0"}, chain),
                        ({"FS_NEAR_FUSED": "0", "FS_LSH_WMAP": "0"}, chain),
                        ({"FS_NEAR_FUSED": "0", "FS_LSH_GRAMTAB": "0"}, chain),
                        ({"FS_SCAN_CAPW": "2"}, fused), ({"FS_LANES": "4"}, fused),
                        # the pending windows eight per wave (k_lsh_batch) on every search, however
                        # few; a wave each (k_lsh_verify) with the Levenshtein distances deferred; and
                        # inside k_lsh_verify
                        ({"FS_LSH_DEFER_MIN": "0"}, fused), ({"FS_LSH_BATCH": "0", "FS_LSH_DEFER_MIN": "0"}, fused),
                        # ... k_lsh_batch walking the buckets of every window instead of enumerating the
                        # script n-grams one slot away
                        ({"FS_LSH_DEFER_MIN": "0", "FS_LSH_EMAP": "0"}, fused),
                        ({"FS_LSH_DEFER_MIN": "0", "FS_LSH_GRAMTAB": "0"}, fused),
                        ({"FS_LSH_DEFER_MIN": "0", "FS_LSH_WMAP": "0"}, fused),
                        ({"FS_LSH_BATCH": "0", "FS_LSH_DEFER_MIN": "1000000000"}, fused)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        full = ScriptIndex(script, swords, emb, normals, cfg=cfg)
        c = full.corpus(tok, off, synth_base["chars"], synth_base["off"])
        assert full.kernel_name(c) == kernel
        got2, st2 = full.search(c)
        assert got.tobytes() == got2.tobytes() and st.matches == st2.matches, env
        got3, _ = full.search(c)
        assert got.tobytes() == got3.tobytes(), env
        full.close()
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("n", [8, 9, 10, 12])
def test_prefilter_scan_at_sub_tile_boundaries(synth_base, monkeypatch, n):
    """k_scan_near8 takes a lane's halo from the next lane and the one after it, and the last
    two lanes of a 512-token sub-tile from the next sub-tile: script spans (verbatim, and with one
    token replaced by its nearest table vector) straddle the sub-tile boundaries in every
    way, and one ends with the corpus' last token.  Records as the
    oracle's, and as the four-tokens-per-lane scan's."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(3000)
    tok, off = util.ragged_corpus([5000, 3, 4100, 2600], script)
    tok = tok.copy()
    cos = emb[script[:2900]] @ emb.T
    cos[np.arange(2900), script[:2900]] = -1.0
    best = cos.argmax(axis=1)
    starts = []
    for j, boundary in enumerate(range(512, len(tok) - 2 * n, 512)):
        at = boundary + 1 - (j % (n + 2))              # the span's first token 1 behind ... n before the boundary
        src = 40 + 31 * j
        w = int(np.searchsorted(off, at, side="right")) - 1
        if at + n > int(off[w + 1]) or at < int(off[w]):   # stays inside its work
            continue
        tok[at:at + n] = script[src:src + n]
        if j % 3 == 1:
            tok[at + j % n] = best[src + j % n]
        starts.append(at % 512)
    assert len(set(starts)) >= n                       # every way of straddling a boundary
    tok[len(tok) - n:] = script[700:700 + n]           # the corpus' last window
    cfg = abi.make_config(window_size=n)
    normals = synth.lsh_normals(n)
    swords = [words[int(t)] for t in script]
    ix, got, st = _run(cfg, script, swords, emb, normals, tok, off,
                       synth_base["chars"], synth_base["off"])
    assert st.path == abi.FS_MODE_GENERAL and len(got) > 100
    assert ix.kernel_name(ix.corpus(tok, off, synth_base["chars"], synth_base["off"])) == "k_near_sift<%d>" % n
    assert int(got["fan_ix"][got["work"] == 3].max()) == 2600 - 1      # the last token is in a record
    for env, kernel in (("FS_SCAN_NEAR8", "k_scan_near<%d>" % n), ("FS_NEAR_FUSED", "k_scan_near8<%d>" % n)):
        monkeypatch.setenv(env, "0")
        old = ScriptIndex(script, swords, emb, normals, cfg=cfg)
        c = old.corpus(tok, off, synth_base["chars"], synth_base["off"])
        assert old.kernel_name(c) == kernel
        got2, st2 = old.search(c)
        assert got.tobytes() == got2.tobytes() and st.matches == st2.matches
        old.close()
        monkeypatch.delenv(env)


def test_general_mode_equals_exact_mode(synth_base):
    """Forcing the LSH pipeline where the proof holds must not change a byte."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(4000)
    tok, off = util.ragged_corpus([900] * 20 + [3, 0, 2100], script)
    normals = synth.lsh_normals(6)
    swords = [words[int(t)] for t in script]
    cfg = abi.make_config(mode=abi.FS_MODE_GENERAL)
    ix, got, st = _run(cfg, script, swords, emb, normals, tok, off,
                       synth_base["chars"], synth_base["off"])
    assert st.path == abi.FS_MODE_GENERAL and ix.info["proof_ok"] == 1
    ex = ScriptIndex(script, swords, emb, normals, cfg=abi.make_config(mode=abi.FS_MODE_EXACT))
    got2, st2 = ex.search(ex.corpus(tok, off, synth_base["chars"], synth_base["off"]))
    assert st2.path == abi.FS_MODE_EXACT
    assert got.tobytes() == got2.tobytes()


@pytest.mark.parametrize("name", util.GOLDEN_CASES)
def test_general_mode_reproduces_golden(name, synth_base):
    from fandom_search_amd.engine import ScriptIndex
    case = util.load_case(name)
    cfg = util.case_config(case, mode=abi.FS_MODE_GENERAL)
    normals = synth.lsh_normals(cfg.window_size, cfg.number_of_hashes, cfg.hash_dimensions)
    words = synth_base["words"]
    script = np.asarray(case["script"], dtype=np.uint32)
    ix = ScriptIndex(script, [words[int(t)] for t in script], synth_base["emb"], normals, cfg=cfg)
    tok, off = util.case_arrays(case)
    rows, st = ix.search(ix.corpus(tok, off, synth_base["chars"], synth_base["off"]))
    assert st.path == abi.FS_MODE_GENERAL
    assert util.rows_to_csv(rows, case, words) == util.golden_text(name, "canonical")


def test_near_duplicate_vectors_give_approximate_matches(synth_base):
    """A table with near-synonyms (cos 0.99): a 6-gram with a synonym swapped in
    is a genuine approximate match (distance ~0.002), found through LSH bucket
    collisions, scored with real dot products."""
    words = synth_base["words"][:2000]
    rng = np.random.default_rng(11)
    emb = synth_base["emb"][:2000].copy()
    for i in range(0, 400, 2):                     # rows i+1 ~ rows i
        v = emb[i] + 0.12 * rng.standard_normal(300).astype(np.float32) / np.sqrt(300)
        emb[i + 1] = v / np.linalg.norm(v)
    script = rng.integers(0, 400, size=1500).astype(np.uint32)
    works = []
    for w in range(10):
        t = rng.integers(400, 2000, size=600).astype(np.uint32)
        for _ in range(4):
            ln = int(rng.integers(6, 20))
            src = int(rng.integers(0, len(script) - ln))
            dst = int(rng.integers(0, 600 - ln))
            span = script[src:src + ln].copy()
            swap = rng.random(ln) < 0.3
            span[swap] ^= 1                         # synonym of each swapped token
            t[dst:dst + ln] = span
        works.append(t)
    tok = np.concatenate(works)
    off = np.arange(11, dtype=np.uint64) * np.uint64(600)
    chars, coff = pack_strings(words)
    cfg = abi.make_config()
    ix, got, st = _run(cfg, script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                       tok, off, chars, coff)
    assert ix.info["proof_ok"] == 0 and ix.info["c_max"] > 0.9
    inexact = got[got["dist"] > 1e-6]
    assert len(inexact) > 10 and float(inexact["dist"].max()) < 0.1


def test_out_of_vocabulary_tokens(synth_base):
    """3-hot vectors (search.py:79-83) in the script and in the fan works."""
    words = list(synth_base["words"][:3000])
    emb = synth_base["emb"][:3000]
    D = 300
    rng = np.random.default_rng(23)

    def oov_id():
        a, b, c = sorted(int(x) for x in rng.integers(0, D, size=3))
        return abi.FS_OOV_FLAG | ((a * D + b) * D + c)

    oov = np.array([oov_id() for _ in range(40)] + [abi.FS_OOV_FLAG | ((7 * D + 7) * D + 9)],
                   dtype=np.uint32)                 # one with a repeated hot position
    strings = words + ["Oov%d" % i for i in range(len(oov))]
    script = rng.integers(0, 3000, size=1200).astype(np.uint32)
    script_str = script.copy()
    for pos in rng.integers(0, 1200, size=60):
        k = int(rng.integers(0, len(oov)))
        script[pos] = oov[k]
        script_str[pos] = 3000 + k
    works, works_str = [], []
    for w in range(8):
        t = rng.integers(0, 3000, size=500).astype(np.uint32)
        ts = t.copy()
        for pos in rng.integers(0, 500, size=15):
            k = int(rng.integers(0, len(oov)))
            t[pos] = oov[k]
            ts[pos] = 3000 + k
        for _ in range(3):
            ln = int(rng.integers(6, 30))
            src = int(rng.integers(0, 1200 - ln))
            dst = int(rng.integers(0, 500 - ln))
            t[dst:dst + ln] = script[src:src + ln]
            ts[dst:dst + ln] = script_str[src:src + ln]
        works.append(t)
        works_str.append(ts)
    tok = np.concatenate(works)
    tok_str = np.concatenate(works_str)
    off = np.arange(9, dtype=np.uint64) * np.uint64(500)
    chars, coff = pack_strings(strings)
    cfg = abi.make_config()
    swords = [strings[int(s)].lower() for s in script_str]
    ix, got, st = _run(cfg, script, swords, emb, synth.lsh_normals(6), tok, off, chars, coff,
                       tok_str=tok_str)
    assert st.path == abi.FS_MODE_GENERAL and len(got) > 0
    hit_oov = [r for r in got if script[r["orig_ix"]] & abi.FS_OOV_FLAG]
    assert hit_oov, "no record landed on an out-of-vocabulary script word"


def test_oov_corpus_on_an_exact_index(synth_base):
    """The index proves the exact scan, the corpus carries OOV ids: that corpus
    alone takes the LSH pipeline."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(2000)
    tok, off = util.ragged_corpus([800] * 6, script)
    tok = tok.copy()
    tok[5::97] = abi.FS_OOV_FLAG | ((1 * 300 + 2) * 300 + 3)
    strings = list(words) + ["Zzz"]
    tok_str = np.where(tok & abi.FS_OOV_FLAG, len(words), tok).astype(np.uint32)
    chars, coff = pack_strings(strings)
    cfg = abi.make_config()
    ix, got, st = _run(cfg, script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                       tok, off, chars, coff, tok_str=tok_str)
    assert ix.info["path"] == abi.FS_MODE_EXACT and st.path == abi.FS_MODE_GENERAL
    clean, off2 = util.ragged_corpus([800] * 6, script)
    got2, st2 = ix.search(ix.corpus(clean, off2, synth_base["chars"], synth_base["off"]))
    assert st2.path == abi.FS_MODE_EXACT and len(got2) >= len(got)


@pytest.mark.parametrize("env", [{}, {"FS_LSH_F32_SLACK": "2e5"}, {"FS_LSH_F32": "0"}, {"FS_LSH_DIAG": "64"}],
                         ids=["f32-sign-path", "forced-f64-fallbacks", "f64-only", "f64-for-oov-windows"])
def test_lsh_key_paths_agree(synth_base, env, monkeypatch):
    """The float32 sign fast path of the LSH keys (with its float64 fallback for
    windows inside the error bound) must give the canonical float64 keys: same rows
    as the oracle with the fast path on, with the bound inflated so that most
    windows fall back, and with the fast path off."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(2500)
    tok, off = util.ragged_corpus([900] * 10 + [0, 7, 2000], script)
    tok = tok.copy()
    tok[11::131] = abi.FS_OOV_FLAG | ((3 * 300 + 4) * 300 + 250)      # windows with an OOV token
    # (round 5: an OOV slot's float32 projection row is the sum of its hot positions' rows --
    # also with two or three of them coinciding, and with two OOV tokens in one window)
    tok[40::173] = abi.FS_OOV_FLAG | ((5 * 300 + 5) * 300 + 9)
    tok[41::173] = abi.FS_OOV_FLAG | ((7 * 300 + 7) * 300 + 7)
    strings = list(words) + ["Oov"]
    tok_str = np.where(tok & abi.FS_OOV_FLAG, len(words), tok).astype(np.uint32)
    chars, coff = pack_strings(strings)
    tok[int(off[3]) + 100:int(off[3]) + 120] = script[500:520]       # a quote with an OOV token inside: records across it
    tok[int(off[3]) + 107] = abi.FS_OOV_FLAG | ((1 * 300 + 2) * 300 + 3)
    tok_str = np.where(tok & abi.FS_OOV_FLAG, len(words), tok).astype(np.uint32)
    cfg = abi.make_config(window_size=8)
    _run(cfg, script, [words[int(t)] for t in script], emb, synth.lsh_normals(8), tok, off,
         chars, coff, tok_str=tok_str)


def test_synonym_rich_table_benchmark_shape(synth_base):
    """The workload of tools/lsh_bench.py (1024 clusters of 8 near-synonyms,
    10 % of the fan tokens swapped for a synonym) against the oracle."""
    emb, perm = synth.clustered_table()
    words = synth_base["words"]
    script = synth.script_tokens(5000)
    tok, off = synth.corpus_tokens(30, 1000, script)
    tok = synth.synonym_swaps(tok, perm)
    ix, got, st = _run(abi.make_config(), script, [words[int(t)] for t in script], emb,
                       synth.lsh_normals(6), tok, off, synth_base["chars"], synth_base["off"])
    assert st.path == abi.FS_MODE_GENERAL and ix.info["c_max"] > 0.9
    assert len(got) > 0 and int((got["dist"] > 0.001).sum()) > 0      # synonyms give approximate matches


@pytest.mark.parametrize("unique", [1, 0])
def test_crowded_buckets_keep_insertion_order(synth_base, unique):
    """A passage that the script repeats 150 times puts 150 windows under one key in every
    table: the CSR buckets (built on the device: count, scan, scatter, sort by window index,
    large buckets by a workgroup) must list them in ascending order, as the reference's
    store_vector does, because NearestFilter keeps the first N of equal distance."""
    words, emb = synth_base["words"], synth_base["emb"]
    rng = np.random.default_rng(99)
    passage = synth.script_tokens(14)
    filler = synth.script_tokens(3000)
    parts = []
    for r in range(150):
        parts.append(passage)
        parts.append(filler[20 * r:20 * r + int(rng.integers(3, 9))])
    script = np.concatenate(parts).astype(np.uint32)
    tok, off = util.ragged_corpus([600] * 6 + [0, 40], script)
    tok = tok.copy()
    tok[int(off[1]) + 50:int(off[1]) + 64] = passage
    tok[int(off[4]) + 300:int(off[4]) + 310] = passage[2:12]
    cfg = abi.make_config(window_size=6, mode=abi.FS_MODE_GENERAL, unique_filter=unique)
    ix, got, st = _run(cfg, script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                       tok, off, synth_base["chars"], synth_base["off"])
    assert st.path == abi.FS_MODE_GENERAL and len(got) > 20


def test_table_with_identical_rows_and_a_zero_row(synth_base):
    """What a real export holds (tools/export_spacy_vectors.py): two ids with the same vector
    (c_max = 1: the exact-n-gram proof fails, a window that swaps one for the other is at
    distance 0 of the script's) and an all-zero row (a window of nothing else has no direction:
    0/0, no match; inside a window it only shortens the vector).  LSH pipeline == oracle."""
    words = synth_base["words"][:600]
    rng = np.random.default_rng(5)
    emb = synth_base["emb"][:600].copy()
    for i in range(0, 60, 2):
        emb[i + 1] = emb[i]                          # ids i and i + 1: the same vector
    emb[100] = 0.0
    emb[101] = 0.0
    script = rng.integers(0, 120, size=900).astype(np.uint32)
    script[50:56] = 100                              # a script window of zero vectors only
    script[200:203] = [100, 7, 101]
    works = []
    for w in range(8):
        t = rng.integers(120, 600, size=500).astype(np.uint32)
        for _ in range(5):
            ln = int(rng.integers(6, 18))
            src = int(rng.integers(0, len(script) - ln))
            dst = int(rng.integers(0, 500 - ln))
            span = script[src:src + ln].copy()
            twin = (span < 60) & (rng.random(ln) < 0.4)
            span[twin] ^= 1                           # the other id of the same vector
            t[dst:dst + ln] = span
        t[490:496] = 100                              # and a fan window of zero vectors only
        works.append(t)
    tok = np.concatenate(works)
    off = np.arange(9, dtype=np.uint64) * np.uint64(500)
    chars, coff = pack_strings(words)
    ix, got, st = _run(abi.make_config(), script, [words[int(t)] for t in script], emb,
                       synth.lsh_normals(6), tok, off, chars, coff)
    assert ix.info["proof_ok"] == 0 and st.path == abi.FS_MODE_GENERAL
    assert len(got) > 50 and not np.isnan(got["dist"]).any()
