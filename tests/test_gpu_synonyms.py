"""Tables with near-synonyms (the shape of every real embedding table): the LSH pipeline
behind the component-id prefilters (fs_lsh.hip: connected components of "near" pairs of
vectors, at most one slot of a neighbour within the threshold joins two components).
Records must equal the oracle's -- which has no shortcut at all -- and the unfiltered LSH
pipeline's byte for byte; a table whose components are too coarse must fall back."""

import numpy as np
import pytest

from fandom_search_amd import abi, synth
from fandom_search_amd.vocab import pack_strings
from tests import util

pytestmark = pytest.mark.gpu


def _clustered(seed=3, clusters=1024, per=8, noise=0.25, scale=None):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((clusters, 300))
    emb = np.repeat(centers, per, axis=0) + noise * rng.standard_normal((clusters * per, 300))
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    if scale is not None:
        emb *= rng.uniform(scale[0], scale[1], size=(len(emb), 1))
    perm = rng.permutation(len(emb))
    out = np.empty_like(emb)
    out[perm] = emb                                    # row perm[i] holds member i % per of cluster i // per
    return np.ascontiguousarray(out, dtype=np.float32), perm, np.argsort(perm)


def _swapped_corpus(script, perm, inv, per, works, tokens, rate=0.1, seed=9):
    tok, off = synth.corpus_tokens(works, tokens, script)
    rng = np.random.default_rng(seed)
    sel = np.nonzero(rng.random(len(tok)) < rate)[0]
    tok[sel] = perm[(inv[tok[sel]] // per) * per + rng.integers(0, per, size=len(sel))].astype(np.uint32)
    return tok, off


def _search(cfg, script, swords, emb, normals, tok, off, chars, coff):
    from fandom_search_amd.engine import ScriptIndex
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    c = ix.corpus(tok, off, chars, coff)
    rows, st = ix.search(c)
    return ix, c, rows, st


@pytest.mark.parametrize("unique", [1, 0])
@pytest.mark.parametrize("n", [6, 7, 8, 10])
def test_component_prefilter_equals_oracle_and_unfiltered_pipeline(synth_base, monkeypatch, n, unique):
    from oracle import c_oracle
    emb, perm, inv = _clustered()
    words = synth_base["words"]
    script = synth.script_tokens(4000)
    swords = [words[int(t)].upper() if i % 11 == 0 else words[int(t)] for i, t in enumerate(script)]
    tok, off = _swapped_corpus(script, perm, inv, 8, 24, 900)
    # spans quoted with a synonym in EVERY slot (no vector id in common with the script), with
    # two unrelated tokens (no match), and verbatim
    rng = np.random.default_rng(5)
    for j in range(6):
        at = int(off[j]) + 300 + 20 * j
        src = 500 + 37 * j
        span = script[src:src + n + 3].copy()
        if j % 3 == 0:
            span = perm[(inv[span] // 8) * 8 + (inv[span] % 8 + 1 + rng.integers(0, 7, size=len(span))) % 8].astype(np.uint32)
        elif j % 3 == 1:
            span[1] = (int(span[1]) + 4001) % len(words)
            span[n - 2] = (int(span[n - 2]) + 1777) % len(words)
        tok[at:at + len(span)] = span
    cfg = abi.make_config(window_size=n, unique_filter=unique)
    normals = synth.lsh_normals(n)
    ix, c, got, st = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
    assert st.path == abi.FS_MODE_GENERAL and ix.info["c_max"] > 0.9
    assert ix.kernel_name(c) == "k_near_sift<%d>" % n                  # the component prefilter ran
    assert 0 < st.candidates < st.windows_processed // 2
    sch, so = pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8)
    want, ost = oi.search(tok, off, synth_base["chars"], synth_base["off"])
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches
    assert int((got["dist"] > 0.001).sum()) > 0                        # approximate matches are records
    # the unfiltered LSH pipeline (keys and buckets for every window): the same bytes; and with the
    # share rule in front of it, which the index takes where the component prefilter is not there
    # (k_share_scan: the windows' keys -- above six slots run by run --, the script windows behind
    # them, the pairs' test, the distance; FS_LSH_SHARE=3: a gate and the pairs' test around the
    # key scan)
    monkeypatch.setenv("FS_LSH_SYN", "0")
    for share, kernel in (("0", "k_lsh_scan"), (None, "k_share_scan<%d>" % n),
                          ("3", "k_lsh_scan")):
        if share is not None:
            monkeypatch.setenv("FS_LSH_SHARE", share)
        ix2, c2, got2, st2 = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
        assert ix2.kernel_name(c2) == kernel
        assert (ix2.share_info()["flags"] != 0) == (share != "0")
        assert got.tobytes() == got2.tobytes() and st.matches == st2.matches, share
        ix2.close()
        monkeypatch.delenv("FS_LSH_SHARE", raising=False)
    monkeypatch.delenv("FS_LSH_SYN")
    # the pending windows a wave each (k_lsh_verify) instead of eight per wave (k_lsh_batch), and
    # k_lsh_batch on every search of an index (the second search of a small batch would take
    # k_lsh_verify: few windows pending): the same bytes
    for env in ({"FS_LSH_BATCH": "0"}, {"FS_LSH_DEFER_MIN": "0"}, {"FS_LSH_DEFER_MIN": "0", "FS_LSH_EMAP": "0"},
                {"FS_LSH_DEFER_MIN": "0", "FS_LSH_GRAMTAB": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ixb, cb, gotb, stb = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
        againb, _ = ixb.search(cb)
        assert got.tobytes() == gotb.tobytes() == againb.tobytes() and st.matches == stb.matches, env
        ixb.close()
        for k in env:
            monkeypatch.delenv(k)
    # round 4's chain (prefilter bitmap, k_expand, k_lsh_sift over every candidate): the same bytes
    monkeypatch.setenv("FS_NEAR_FUSED", "0")
    ix3, c3, got3, st3 = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
    assert ix3.kernel_name(c3) == ("k_scan_near<6>" if n == 6 else "k_scan_near8<%d>" % n)
    assert got.tobytes() == got3.tobytes() and st.matches == st3.matches


def test_vectors_of_different_length(synth_base, monkeypatch):
    """Norms between 0.8 and 1.25: the line a pair must be above to be near depends on the
    product of its norms; records as the oracle's."""
    from oracle import c_oracle
    emb, perm, inv = _clustered(seed=11, clusters=512, per=16, noise=0.2, scale=(0.8, 1.25))
    words = synth_base["words"]
    script = synth.script_tokens(3000)
    swords = [words[int(t)] for t in script]
    tok, off = _swapped_corpus(script, perm, inv, 16, 20, 800, rate=0.15)
    cfg = abi.make_config()
    normals = synth.lsh_normals(6)
    ix, c, got, st = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
    assert ix.kernel_name(c) == "k_near_sift<6>"
    sch, so = pack_strings(swords)
    want, ost = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8).search(
        tok, off, synth_base["chars"], synth_base["off"])
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and len(got) > 0


def test_coarse_components_fall_back(synth_base, monkeypatch):
    """Sixteen clusters of 512 near-synonyms: a component holds a sixteenth... of the table
    each, and a zero row joins every component it touches -- more than an eighth of the table
    in one component: no prefilter, the plain LSH pipeline, the oracle's records."""
    from oracle import c_oracle
    emb, perm, inv = _clustered(seed=7, clusters=16, per=512, noise=0.2)
    emb[perm[5]] = 0.0                                # a zero row: near everything
    words = synth_base["words"]
    script = synth.script_tokens(2000)
    script[100] = perm[5]                             # ... and it occurs in the script
    swords = [words[int(t)] for t in script]
    tok, off = _swapped_corpus(script, perm, inv, 512, 10, 500)
    cfg = abi.make_config()
    normals = synth.lsh_normals(6)
    ix, c, got, st = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
    assert st.path == abi.FS_MODE_GENERAL
    assert ix.kernel_name(c) == "k_share_scan<6>"     # fell back: no component prefilter -- the share rule then
    sch, so = pack_strings(swords)
    want, ost = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8).search(
        tok, off, synth_base["chars"], synth_base["off"])
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches
    monkeypatch.setenv("FS_LSH_SHARE", "0")           # ... and the key scan by itself
    ix2, c2, got2, st2 = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
    assert ix2.kernel_name(c2) == "k_lsh_scan" and got2.tobytes() == got.tobytes()


def test_largest_filter_at_n6_leaves_the_second_filter_out(synth_base, monkeypatch):
    """A 3-gram filter of 2^15 words (scripts with more than ~22k distinct 6-grams) and the
    64 KiB middle-slot filter of k_scan_near<6> together exceed the CU's 160 KiB of LDS: the
    launch must leave the optional second filter out, not fail."""
    from oracle import c_oracle
    monkeypatch.setenv("FS_FILTER_LOG2_WORDS", "15")
    monkeypatch.setenv("FS_NEAR_FUSED", "0")            # (k_near_sift's filter has 2^14 words at most)
    emb, perm, inv = _clustered()
    words = synth_base["words"]
    script = synth.script_tokens(3000)
    swords = [words[int(t)] for t in script]
    tok, off = _swapped_corpus(script, perm, inv, 8, 16, 700)
    cfg = abi.make_config()
    normals = synth.lsh_normals(6)
    ix, c, got, st = _search(cfg, script, swords, emb, normals, tok, off, synth_base["chars"], synth_base["off"])
    assert ix.kernel_name(c) == "k_scan_near<6>"
    sch, so = pack_strings(swords)
    want, ost = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8).search(
        tok, off, synth_base["chars"], synth_base["off"])
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and len(got) > 0
