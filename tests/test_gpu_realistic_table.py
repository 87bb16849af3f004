"""A table with the features of a real embedding table (synth.realistic_table: unnormalised
vectors, similarity at three scales, duplicate and zero rows) under fan text with
out-of-vocabulary names and capitalised words: records as the C oracle's, byte for byte, on
more than fifty works, with and without the out-of-vocabulary tokens (which decide the
pipeline: a batch that holds any takes the plain LSH pipeline)."""

import numpy as np
import pytest

from fandom_search_amd import abi, synth
from fandom_search_amd.vocab import pack_strings
from tests import util

pytestmark = pytest.mark.gpu

ROWS = 6000


@pytest.fixture(scope="module")
def real():
    emb, group = synth.realistic_table(rows=ROWS)
    strings, vid = synth.realistic_vector_ids(ROWS)
    script = synth._draw(np.random.default_rng(77), 3000, ROWS)
    return emb, group, strings, vid, script


@pytest.mark.parametrize("oov", [True, False])
@pytest.mark.parametrize("unique", [0, 1])
def test_realistic_table_equals_oracle(real, oov, unique, monkeypatch):
    from oracle import c_oracle
    from fandom_search_amd.engine import ScriptIndex
    emb, group, strings, vid, script = real
    tok_str, off = synth.realistic_corpus(60, 700, script, group, ROWS, oov_rate=0.08 if oov else 0.0)
    # planted quotes that are sure to be there: verbatim, with near-synonyms, with a name inside
    rng = np.random.default_rng(3)
    for j in range(12):
        at = int(off[j]) + 50 + 7 * j
        span = script[200 + 40 * j:200 + 40 * j + 14].astype(np.uint32)
        tok_str[at:at + len(span)] = span
        if j % 3 == 1:
            k = at + 3 + j % 5
            mates = np.nonzero(group[:, 2] == group[int(tok_str[k]), 2])[0]
            tok_str[k] = mates[int(rng.integers(0, len(mates)))]
        if j % 3 == 2 and oov:
            tok_str[at + 6] = 2 * ROWS + j
    tok_vec = vid[tok_str]
    swords = [strings[int(t)].upper() if i % 9 == 0 else strings[int(t)] for i, t in enumerate(script)]
    chars, coff = pack_strings(strings)
    cfg = abi.make_config(unique_filter=unique)
    normals = synth.lsh_normals(6)
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    c = ix.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
    got, st = ix.search(c)
    again, _ = ix.search(c)
    assert st.path == abi.FS_MODE_GENERAL and got.tobytes() == again.tobytes()
    # neither integer prefilter applies (norms from 1.8 to 18): the share rule in front of the LSH work
    assert ix.kernel_name(c) == "k_share_scan<6>" and ix.share_info()["flags"] & 32
    info = ix.share_info()
    sizes, used = ix.component_sizes()                 # (the rule's components: the nested groups, not one giant one)
    assert used and len(sizes) == info["components"] and int(sizes.max()) == info["largest"] < ROWS // 50
    assert int(sizes.sum()) == ROWS and abs(info["gamma"] - 0.7) < 1e-12
    assert ix.share_counts()["windows"] == 0           # (counters are off unless FS_SHARE_COUNT is set)
    sch, so = pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8)
    want, ost = oi.search(tok_vec, off, chars, coff, tok_str=tok_str)
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and len(got) > 100
    assert int((got["dist"] > 1e-3).sum()) > 0            # near-synonym matches are records
    oi.close()
    ix.close()
    # the share rule's other forms, and none: the same bytes.  0: the key scan over every window;
    # 3: the gate and the pairs' test inside the key scan; 1, 2: either by itself; 7: the gate over
    # the subsets that are heavy on both sides; 43, 11: out-of-vocabulary fan tokens as possibly
    # near anything (what an index does whose table does not prove them far)
    for share, kernel in (("0", "k_lsh_scan"), ("3", "k_lsh_scan"), ("1", "k_lsh_scan"), ("2", "k_lsh_scan"),
                          ("7", "k_lsh_scan"), ("43", "k_share_scan<6>"), ("11", "k_lsh_scan")):
        monkeypatch.setenv("FS_LSH_SHARE", share)
        ix2 = ScriptIndex(script, swords, emb, normals, cfg=cfg)
        c2 = ix2.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
        got2, st2 = ix2.search(c2)
        assert ix2.kernel_name(c2) == kernel
        assert got2.tobytes() == got.tobytes() and st2.matches == st.matches, share
        ix2.close()


def test_script_with_names_of_its_own(real, monkeypatch):
    """A script whose tokens include out-of-vocabulary names (what every real script does): the
    share rule still applies -- the names' 3-hot vectors get components of their own, by their
    sets of hot positions -- and the cases its analysis singles out are all in the text: a name
    whose three hashes fell on two positions, the same set under another id, a name that contains
    it, a one-position name and a two-position name over it, a two-position subset of a
    three-position name of the script.  Records as the C oracle's and as the key scan's."""
    from oracle import c_oracle
    from fandom_search_amd.engine import ScriptIndex
    emb, group, strings, vid, script = real
    D, F = 300, abi.FS_OOV_FLAG
    code = lambda a, b, c: np.uint32(F | ((a * D + b) * D + c))
    h = lambda i: [int(x) for x in ((int(vid[2 * ROWS + i]) & ~F) // (D * D), ((int(vid[2 * ROWS + i]) & ~F) // D) % D, (int(vid[2 * ROWS + i]) & ~F) % D)]
    x, y, z = h(0)                                    # a regular name of the script: three positions
    assert x < y < z
    extra = [("Deg0", code(7, 7, 19)), ("Deg1", code(7, 19, 19)), ("Deg2", code(7, 19, 123)), ("One0", code(55, 55, 55)),
             ("Two0", code(55, 55, 201)), ("Sub0", code(x, y, y)), ("Sub1", code(x, x, z))]
    strings = list(strings) + [w for w, _ in extra]
    vid = np.concatenate([vid, np.array([v for _, v in extra], dtype=np.uint32)])
    sid = {w: len(strings) - len(extra) + i for i, (w, _) in enumerate(extra)}
    name = lambda i: 2 * ROWS + i
    rng = np.random.default_rng(21)
    script_str = script.astype(np.uint32).copy()
    for i in range(0, len(script_str), 9):            # a name every ninth token, thirty of them in turn
        script_str[i] = name((i // 9) % 30)
    script_str[100], script_str[103] = sid["Deg0"], sid["One0"]
    script_str[1000], script_str[1002] = sid["Deg0"], name(0)
    script_vec = vid[script_str]
    tok_str, off = synth.realistic_corpus(50, 600, script, group, ROWS, oov_rate=0.08)
    for j in range(20):                               # quotes, names and all; the singled-out ones varied
        at = int(off[j]) + 40 + 11 * j
        src = 94 if j % 2 else 994
        span = script_str[src:src + 14].copy()
        swap = {sid["Deg0"]: [sid["Deg0"], sid["Deg1"], sid["Deg2"], name(44)][j % 4],
                sid["One0"]: [sid["One0"], sid["Two0"]][(j // 2) % 2],
                name(0): [name(0), sid["Sub0"], sid["Sub1"], name(45)][(j // 2) % 4]}
        tok_str[at:at + 14] = [swap.get(int(t), int(t)) for t in span]
    tok_vec = vid[tok_str]
    swords = [strings[int(t)] for t in script_str]
    chars, coff = pack_strings(strings)
    cfg = abi.make_config()
    normals = synth.lsh_normals(6)
    ix = ScriptIndex(script_vec, swords, emb, normals, cfg=cfg)
    c = ix.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
    got, st = ix.search(c)
    assert ix.kernel_name(c) == "k_share_scan<6>" and ix.share_info()["flags"] & 32
    sch, so = pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script_vec, sch, so, emb, normals, threads=8)
    want, ost = oi.search(tok_vec, off, chars, coff, tok_str=tok_str)
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and len(got) > 200
    oi.close()
    ix.close()
    # ... and what passes what, from an index that counts: every window through the filter or not,
    # the flagged ones a superset of the windows with records
    monkeypatch.setenv("FS_SHARE_COUNT", "1")
    ixc = ScriptIndex(script_vec, swords, emb, normals, cfg=cfg)
    monkeypatch.delenv("FS_SHARE_COUNT")
    cc = ixc.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
    gotc, stc = ixc.search(cc)
    n = ixc.share_counts()
    assert gotc.tobytes() == got.tobytes()
    # (the kernel sees the token stream: the windows across two works are dropped behind it)
    assert n["windows"] == len(tok_vec) - 5 >= stc.windows_processed
    assert 0 < n["windows_with_a_key_in_the_filter"] < n["windows"]
    assert n["windows_flagged"] >= stc.candidates > 0 and n["pairs_tested"] >= n["distances"] >= n["windows_flagged"] - n["windows_flagged_as_they_are"]
    assert len(np.unique(got["work"].astype(np.int64) << 32 | got["fan_ix"])) > 0
    ixc.close()
    for share in ("0", "3"):
        monkeypatch.setenv("FS_LSH_SHARE", share)
        ix2 = ScriptIndex(script_vec, swords, emb, normals, cfg=cfg)
        c2 = ix2.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
        got2, st2 = ix2.search(c2)
        assert ix2.kernel_name(c2) == "k_lsh_scan" and got2.tobytes() == got.tobytes(), share
        ix2.close()


def test_ragged_batches_under_the_share_rule(real):
    """Empty works, works shorter than a window, a work of exactly one window, a batch of one
    token and an empty batch through k_share_scan: records as the C oracle's (the kernel sees the
    token stream -- windows across two works must not become records)."""
    from oracle import c_oracle
    from fandom_search_amd.engine import ScriptIndex
    emb, group, strings, vid, script = real
    tok_str, off = synth.realistic_corpus(12, 300, script, group, ROWS, oov_rate=0.08)
    quote = script[500:520].astype(np.uint32)
    lengths = [0, 3, 6, 300, 0, 0, 5, 7, 250, 1, 6, 20, 0]
    parts, at = [], 0
    for i, ln in enumerate(lengths):
        p = tok_str[at:at + ln].copy()
        at += 300
        if ln >= 6:
            p[:min(ln, 20)] = quote[:min(ln, 20)]       # every work that can hold a window holds a quote
        parts.append(p)
    tok_s = np.concatenate(parts).astype(np.uint32)
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lengths)
    tok_v = vid[tok_s]
    swords = [strings[int(t)] for t in script]
    chars, coff = pack_strings(strings)
    cfg = abi.make_config()
    normals = synth.lsh_normals(6)
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    sch, so = pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=4)
    for tv, ts, o in ((tok_v, tok_s, off), (tok_v[:1], tok_s[:1], np.array([0, 1], dtype=np.uint64)),
                      (tok_v[:0], tok_s[:0], np.array([0], dtype=np.uint64)),
                      (tok_v[:0], tok_s[:0], np.array([0, 0, 0], dtype=np.uint64))):
        c = ix.corpus(tv, o, chars, coff, tok_str=ts)
        got, st = ix.search(c)
        want, ost = oi.search(tv, o, chars, coff, tok_str=ts)
        util.assert_rows_equal(got, want)
        assert st.matches == ost.matches and st.windows_processed == ost.windows_processed
        if len(tv) > 1:
            assert ix.kernel_name(c) == "k_share_scan<6>" and len(got) > 50
        c.close()
    oi.close()
    ix.close()
