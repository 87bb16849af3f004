"""GPU parity: rows of the HIP library (through the C ABI) against the plain-C
oracle (reference algorithm, canonical arithmetic) on the same seeded inputs.
Integer fields and both float64 fields must be bit-identical."""

import os

import numpy as np
import pytest

from fandom_search_amd import abi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _case(synth_base, lengths, script_tokens, n=6, first_work=0, **env):
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    normals = synth.lsh_normals(n)
    script = synth.script_tokens(script_tokens)
    tok, off = util.ragged_corpus(lengths, script, first_work)
    cfg = abi.make_config(window_size=n)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        ix = ScriptIndex(script, [words[int(t)] for t in script], emb, normals, cfg=cfg)
        corpus = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
        got, st = ix.search(corpus)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    oi = util.oracle_index(cfg, script, words, emb, normals)
    want, ost = oi.search(tok, off, synth_base["chars"], synth_base["off"])
    util.assert_rows_equal(got, want)
    assert st.windows_processed == ost.windows_processed
    assert st.matches == ost.matches
    assert st.rows == len(want)
    assert ix.info["proof_ok"] == 1 and ix.info["path"] == abi.FS_MODE_EXACT
    corpus.close()
    ix.close()
    oi.close()
    return got, st


def test_c1_shape(synth_base):
    """BASELINE.json configs[0]: 50 works x 1000 tokens vs a 500-line script."""
    got, st = _case(synth_base, [1000] * 50, 5000)
    assert len(got) > 0


def test_ragged_and_empty_works(synth_base):
    lengths = [0, 3, 5, 6, 7, 0, 0, 255, 256, 257, 1023, 1024, 1025, 1, 2000, 0]
    _case(synth_base, lengths, 3000)


def test_single_short_and_no_works(synth_base):
    _case(synth_base, [4], 1000)
    _case(synth_base, [], 1000)
    _case(synth_base, [0, 0], 1000)


@pytest.mark.parametrize("tpl", [4, 8])
@pytest.mark.parametrize("n", [2, 5, 6])
def test_scan_tokens_per_lane_layouts(synth_base, tpl, n):
    """Both bitmap layouts of the chained kernels' scan: 8 tokens per lane (k_scan8) and 4
    (k_scan_simple)."""
    _case(synth_base, [1500] * 20 + [0, 511, 512, 513, 1, 4000], 4000, n=n, FS_SCAN_TPL=tpl)


def test_scan_simple_variant(synth_base):
    _case(synth_base, [1500] * 40, 4000, FS_SCAN_VARIANT="simple")


@pytest.mark.parametrize("lw", [10, 12, 15])
def test_filter_sizes(synth_base, lw):
    _case(synth_base, [2000] * 30, 8000, FS_FILTER_LOG2_WORDS=lw, FS_SFILTER_LOG2_WORDS=lw)
    _case(synth_base, [2000] * 30, 8000, FS_FILTER_LOG2_WORDS=lw, FS_SFILTER_LOG2_WORDS=lw, FS_LANES=2)


def test_window_size_4(synth_base):
    _case(synth_base, [800] * 30, 3000, n=4)


def test_c2_slice(synth_base):
    """600 works of BASELINE.json configs[1] (2000 tokens, 20k-token script)."""
    got, st = _case(synth_base, [2000] * 600, 20000)
    assert st.scan_ms > 0


@pytest.mark.parametrize("plain", [0.34, 0.85])
def test_separate_string_ids(synth_base, monkeypatch, plain):
    """Fan tokens whose text differs from the text of their vector row (the
    reference's fan side is case-sensitive, the script side lower-cased:
    search.py:151 vs :166): string ids travel next to vector ids and the
    Levenshtein distance is computed per match -- except for the matches whose tokens
    all carry string id == vector id, which take it from the per-gram table
    (FS_STR_LEVTAB=0: none do; same bytes).  `plain`: share of such tokens."""
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings
    words, emb = synth_base["words"], synth_base["emb"]
    n = 6
    normals = synth.lsh_normals(n)
    script = synth.script_tokens(3000)
    tok, off = util.ragged_corpus([700] * 25 + [0, 9, 1300], script)
    rng = np.random.default_rng(5)
    # string table: lower-case words, then Capitalised, then UPPER + '!!'
    strings = list(words) + [w.capitalize() for w in words] + [w.upper() + "!!" for w in words]
    rest = (1.0 - plain) / 2
    variant = rng.choice(3, size=len(tok), p=[plain, rest, rest]).astype(np.uint32)
    tok_str = (tok + variant * len(words)).astype(np.uint32)
    chars, coff = pack_strings(strings)
    cfg = abi.make_config(window_size=n)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, normals, cfg=cfg)
    got, st = ix.search(ix.corpus(tok, off, chars, coff, tok_str=tok_str))
    oi = util.oracle_index(cfg, script, words, emb, normals)
    want, ost = oi.search(tok, off, chars, coff, tok_str=tok_str)
    util.assert_rows_equal(got, want)
    assert len(set(got["lev"].tolist())) > 3      # text variants change the distance
    assert st.matches == ost.matches
    monkeypatch.setenv("FS_STR_LEVTAB", "0")
    ix2 = ScriptIndex(script, [words[int(t)] for t in script], emb, normals, cfg=cfg)
    got2, st2 = ix2.search(ix2.corpus(tok, off, chars, coff, tok_str=tok_str))
    assert got.tobytes() == got2.tobytes() and st.matches == st2.matches
    ix2.close()


def test_string_ids_lane_levenshtein(synth_base, monkeypatch):
    """lev_lane (a lane per hit: Myers' recurrence over character classes of the script's
    alphabet; inside k_scan_rows' rounds, and in k_strbest of the chained kernels) against the
    oracle where its special cases meet: fan words of more than 15 code
    points, code points outside the script's alphabet and outside ASCII, script windows of
    more than 64 code points (scratch DP), n-grams that occur several times in the script
    with different text (several ranks per hit); all three forms give the same bytes."""
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings
    words, emb = synth_base["words"], synth_base["emb"]
    n = 6
    normals = synth.lsh_normals(n)
    script = synth.script_tokens(3000).copy()
    script[400:440] = script[40:80]                       # the same n-grams again ...
    swords = [words[int(t)] for t in script]
    for i in range(400, 440, 3):
        swords[i] = swords[i].upper()                      # ... with other text: ranks differ
    for i in range(900, 960):
        swords[i] = swords[i] * 4                          # windows of more than 64 code points
    for i in range(1500, 1510):
        swords[i] = swords[i] + "\u00e9\u4e16"            # non-ASCII script characters
    tok, off = util.ragged_corpus([700] * 25 + [0, 9, 1300], script)
    rng = np.random.default_rng(17)
    V = len(words)
    strings = (list(words) + [w.capitalize() for w in words] + [w * 5 for w in words] +
               [w + "\u00e9" for w in words] + [w[:1] + "\U0001F600" + w[1:] for w in words])
    variant = rng.choice(5, size=len(tok), p=[0.6, 0.1, 0.1, 0.1, 0.1]).astype(np.uint32)
    tok_str = (tok + variant * V).astype(np.uint32)
    chars, coff = pack_strings(strings)
    cfg = abi.make_config(window_size=n)
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    got, st = ix.search(ix.corpus(tok, off, chars, coff, tok_str=tok_str))
    oi = util.oracle_index(cfg, script, words, emb, normals, swords=swords)
    want, ost = oi.search(tok, off, chars, coff, tok_str=tok_str)
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and len(set(got["lev"].tolist())) > 6
    c0 = ix.corpus(tok, off, chars, coff, tok_str=tok_str)
    assert ix.kernel_name(c0) == "k_scan_rows<6,4>"        # per-hit Levenshtein inside the rounds
    # the chained kernels with k_strbest (a lane per hit), and with the wave-per-pair kernels
    for env in ("FS_STR_FUSED", "FS_STR_FAST"):
        monkeypatch.setenv(env, "0")
        ix2 = ScriptIndex(script, swords, emb, normals, cfg=cfg)
        c2 = ix2.corpus(tok, off, chars, coff, tok_str=tok_str)
        assert ix2.kernel_name(c2) == "k_scan8<6>"
        got2, st2 = ix2.search(c2)
        assert got.tobytes() == got2.tobytes() and st.matches == st2.matches
        ix2.close()
        monkeypatch.delenv(env)


def test_packed_wire_rows_round_trip(synth_base):
    """16-byte wire records of the exact pipeline expand to the same fs_row bytes."""
    import torch
    from fandom_search_amd import _lib
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(6000)
    tok, off = util.ragged_corpus([1200] * 40 + [0, 5, 3000], script)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    corpus = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    want, _ = ix.search(corpus)
    cap = len(want) + 10
    packed = torch.zeros(cap * 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # (as below: the fill is torch's, the search writes on the library's streams)
    n, st = ix.search_device(corpus, packed.data_ptr(), cap, packed=True)
    assert n == len(want)
    full = torch.empty(cap * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # torch's fill / copy is complete before the library's own streams touch the buffer (engine.torch_ready)
    ix.unpack_device(packed.data_ptr(), n, full.data_ptr())
    got = full.cpu().numpy()[:n * 32].view(abi.ROW_DTYPE)
    assert got.tobytes() == want.tobytes()
    with pytest.raises(_lib.FsError) as e:
        ix.search_device(corpus, packed.data_ptr(), 10, packed=True)
    assert e.value.code == abi.FS_E_CAPACITY and e.value.required == len(want)
    # the LSH pipeline has no packed form
    gx = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config(mode=abi.FS_MODE_GENERAL))
    gc = gx.corpus(tok, off, synth_base["chars"], synth_base["off"])
    with pytest.raises(_lib.FsError) as e:
        gx.search_device(gc, packed.data_ptr(), cap, packed=True)
    assert e.value.code == abi.FS_E_UNSUPPORTED


def test_long_tokens_and_levenshtein_limit(synth_base, monkeypatch):
    """Fan texts of any length are computed on the device by the lane-per-hit Levenshtein of
    batches with string ids (bit-equal to the oracle); the wave-per-pair kernels behind
    FS_STR_FAST=0 hold their operands in LDS: up to 512 code points per side, longer n-gram
    texts are refused loudly."""
    from fandom_search_amd import _lib
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings
    words, emb = synth_base["words"], synth_base["emb"]
    n = 6
    script = synth.script_tokens(800)
    tok, off = util.ragged_corpus([400] * 6, script)
    # fan strings: every third word id gets a 70-character spelling (6 of them: 6*70+12 = 432)
    strings = [w if i % 3 else (w * 18)[:70].upper() for i, w in enumerate(words)]
    chars, coff = pack_strings(strings)
    cfg = abi.make_config(window_size=n)
    normals = synth.lsh_normals(n)
    swords = [words[int(t)] for t in script]
    too_long = [w * 30 for w in words]                # 120 code points per word: 6*120+12 > 512
    chars2, coff2 = pack_strings(too_long)
    oi = util.oracle_index(cfg, script, words, emb, normals)
    want, _ = oi.search(tok, off, chars, coff, tok_str=tok)
    want2, _ = oi.search(tok, off, chars2, coff2, tok_str=tok)
    for fast in ("1", "0"):
        monkeypatch.setenv("FS_STR_FAST", fast)
        ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
        got, st = ix.search(ix.corpus(tok, off, chars, coff, tok_str=tok))
        util.assert_rows_equal(got, want)
        assert int(got["lev"].max()) > 200
        if fast == "1":
            got2, _ = ix.search(ix.corpus(tok, off, chars2, coff2, tok_str=tok))
            util.assert_rows_equal(got2, want2)
            assert int(got2["lev"].max()) > 512
            # ... up to what the 8-byte wire records hold (a distance below 1024)
            chars3, coff3 = pack_strings([w * 70 for w in words])
            with pytest.raises(_lib.FsError) as e:
                ix.search(ix.corpus(tok, off, chars3, coff3, tok_str=tok))
            assert e.value.code == abi.FS_E_UNSUPPORTED
        else:
            with pytest.raises(_lib.FsError) as e:
                ix.search(ix.corpus(tok, off, chars2, coff2, tok_str=tok))
            assert e.value.code == abi.FS_E_UNSUPPORTED and "512" in str(e.value)
        ix.close()


def test_host_rows_stored_by_the_search_or_copied(synth_base, monkeypatch):
    """Host rows of the exact pipeline: 8-byte records stored into pinned host memory by the
    search's last kernel (default), copied after the search (FS_HOST_ZEROCOPY=0) or copied as
    32-byte rows (FS_HOST_WIRE8=0): the same bytes, also with several lanes (k_compact stores
    them) and into a caller's buffer that is too small at first."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(4000)
    tok, off = util.ragged_corpus([900] * 40 + [0, 5, 2500], script)
    cfg = abi.make_config()
    swords = [words[int(t)] for t in script]
    want = None
    for env in ({}, {"FS_HOST_ZEROCOPY": "0"}, {"FS_HOST_WIRE8": "0"}, {"FS_LANES": "4"},
                {"FS_LANES": "4", "FS_HOST_ZEROCOPY": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ix = ScriptIndex(script, swords, emb, synth.lsh_normals(6), cfg=cfg)
        c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
        got, st = ix.search(c, cap=16)                   # (grows to what the search reports)
        again, _ = ix.search(c, reuse=True)
        assert len(got) > 500 and got.tobytes() == again.tobytes()
        want = got.tobytes() if want is None else want
        assert got.tobytes() == want, env
        ix.close()
        for k in env:
            monkeypatch.delenv(k)


def test_searches_in_flight(synth_base):
    """fs_search_corpus_begin/_end: several searches queued before the first is
    collected give the rows of the synchronous call."""
    import torch
    from fandom_search_amd import _lib
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(4000)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    corpora, want = [], []
    for k in range(4):
        tok, off = util.ragged_corpus([600 + 100 * k] * (10 + k) + [0, 4], script, first_work=50 * k)
        c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
        corpora.append(c)
        want.append(ix.search(c)[0])
    bufs = [torch.zeros((len(w) + 8) * 32, dtype=torch.uint8, device="cuda") for w in want]
    torch.cuda.synchronize()      # torch's fill / copy is complete before the library's own streams touch the buffer (engine.torch_ready)
    tickets = [ix.search_begin(c, b.data_ptr(), len(w) + 8) for c, b, w in zip(corpora, bufs, want)]
    with pytest.raises(_lib.FsError, match="in flight"):
        ix.search_begin(corpora[0], bufs[0].data_ptr(), 8)
    for order in (2, 0, 3, 1):                       # collected out of order
        n, st = ix.search_end(tickets[order])
        got = bufs[order].cpu().numpy()[:n * 32].view(abi.ROW_DTYPE)
        assert got.tobytes() == want[order].tobytes()
        assert st.scan_ms > 0
    with pytest.raises(_lib.FsError, match="no such search"):
        ix.search_end(tickets[0])
    # a too small buffer is reported at _end, and the slot is free again afterwards
    t = ix.search_begin(corpora[1], bufs[1].data_ptr(), 3)
    with pytest.raises(_lib.FsError) as e:
        ix.search_end(t)
    assert e.value.code == abi.FS_E_CAPACITY and e.value.required == len(want[1])
    assert ix.search(corpora[1])[0].tobytes() == want[1].tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16])
def test_window_sizes_on_a_one_hot_table(n):
    """Orthogonal vectors (c_max = 0) and threshold 0.05: one substituted token
    gives cos = (n-1)/n <= 0.9375 < 0.95, so the exact-scan proof holds for every
    window size up to 16, including those without a specialised scan kernel
    (n = 1, 9, 11, 16 use the generic one).  (At the default threshold 0.1 the proof
    rightly fails from n = 10 on: 10/11 > 0.9 is a genuine approximate match.)"""
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings
    from oracle import c_oracle
    rng = np.random.default_rng(n)
    V, D = 40, 48
    emb = np.zeros((V, D), dtype=np.float32)
    emb[np.arange(V), rng.permutation(D)[:V]] = 1.0
    words = ["t%d" % i for i in range(V)]
    script = rng.integers(0, V, size=900).astype(np.uint32)
    works = []
    for w in range(12):
        t = rng.integers(0, V, size=700).astype(np.uint32)
        for _ in range(5):
            ln = int(rng.integers(n, 3 * n + 2))
            src = int(rng.integers(0, len(script) - ln))
            dst = int(rng.integers(0, 700 - ln))
            t[dst:dst + ln] = script[src:src + ln]
        works.append(t)
    works += [np.zeros(0, np.uint32), script[:n].copy(), script[5:5 + max(n - 1, 0)].copy()]
    off = np.zeros(len(works) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(w) for w in works])
    tok = np.concatenate(works)
    chars, coff = pack_strings(words)
    cfg = abi.make_config(window_size=n, emb_dim=D, number_of_hashes=4, hash_dimensions=6,
                          distance_threshold=0.05)
    normals = rng.standard_normal((4, 6, D * n))
    swords = [words[int(t)] for t in script]
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    assert ix.info["proof_ok"] == 1 and ix.info["path"] == abi.FS_MODE_EXACT
    got, st = ix.search(ix.corpus(tok, off, chars, coff))
    sch, so = pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8)
    want, ost = oi.search(tok, off, chars, coff)
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and len(got) > 0
    assert st.path == abi.FS_MODE_EXACT


def test_one_very_long_work(synth_base):
    """A single work of 3M tokens (many tiles, one work offset pair)."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(3000)
    rng = np.random.default_rng(77)
    tok = synth._draw(rng, 3_000_000, len(words))
    for _ in range(3000):
        ln = int(rng.integers(6, 25))
        src = int(rng.integers(0, len(script) - ln))
        dst = int(rng.integers(0, len(tok) - ln))
        tok[dst:dst + ln] = script[src:src + ln]
    off = np.array([0, len(tok)], dtype=np.uint64)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    got, st = ix.search(ix.corpus(tok, off, synth_base["chars"], synth_base["off"]))
    assert st.windows_processed == len(tok) - 5
    # oracle on the first 60k tokens as its own work: the rows that lie fully inside agree
    cut = 60_000
    oi = util.oracle_index(abi.make_config(), script, words, emb, synth.lsh_normals(6))
    want, _ = oi.search(tok[:cut], np.array([0, cut], np.uint64), synth_base["chars"],
                        synth_base["off"])
    inside = got[got["fan_ix"] < cut - 12]
    util.assert_rows_equal(inside, want[want["fan_ix"] < cut - 12])
    assert np.all(np.diff(got["fan_ix"].astype(np.int64)) > 0) and len(got) > 5000


@pytest.mark.parametrize("lanes", [1, 2, 4])
@pytest.mark.parametrize("mode", [abi.FS_MODE_AUTO, abi.FS_MODE_GENERAL])
def test_overlapping_searches_stress(synth_base, mode, lanes, monkeypatch):
    """With FS_LANES > 1 searches alternate between the index's lanes (streams with
    their own workspaces), so consecutive ones run side by side on the GPU.  Many
    rounds of four searches in flight over corpora of different sizes, collected
    in changing order, must reproduce the synchronous rows every time."""
    import torch
    from fandom_search_amd.engine import ScriptIndex
    monkeypatch.setenv("FS_LANES", str(lanes))
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(6000)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config(mode=mode))
    sizes = [(400, 2000), (40, 700), (900, 1500), (150, 3000)] if mode == abi.FS_MODE_AUTO else \
            [(60, 900), (10, 400), (90, 700), (30, 1200)]
    corpora, want = [], []
    for k, (works, tokens) in enumerate(sizes):
        tok, off = synth.corpus_tokens(works, tokens, script, first_work=1000 * k)
        c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
        corpora.append(c)
        want.append(ix.search(c)[0])
    bufs = [torch.zeros((len(w) + 8) * 32, dtype=torch.uint8, device="cuda") for w in want]
    torch.cuda.synchronize()      # torch's fill / copy is complete before the library's own streams touch the buffer (engine.torch_ready)
    rounds = 25 if mode == abi.FS_MODE_AUTO else 6
    for r in range(rounds):
        order = [(r + j) % 4 for j in range(4)]
        for b in bufs:
            b.zero_()
        tickets = {k: ix.search_begin(corpora[k], bufs[k].data_ptr(), len(want[k]) + 8) for k in order}
        for k in reversed(order) if r % 2 else order:
            n, st = ix.search_end(tickets[k])
            assert n == len(want[k])
            got = bufs[k].cpu().numpy()[:n * 32].view(abi.ROW_DTYPE)
            assert got.tobytes() == want[k].tobytes(), (r, k)
    ix.close()


def test_table_overflow_buckets_with_colliding_hashes(synth_base):
    """The verification table is hash-and-displace; n-grams whose 32-bit hashes
    are identical cannot be separated by a displacement seed and fall back to
    linear probing inside an overflow bucket.  A long script of 2-grams holds
    such pairs (checked here with the hash restated in numpy); rows must still
    equal the oracle's."""
    words, emb = synth_base["words"], synth_base["emb"]
    rng = np.random.default_rng(5)
    script = rng.integers(0, len(words), size=120_000).astype(np.uint32)
    m = (script.astype(np.uint64) * 0x9E3779) & 0xFFFFFFFF
    x = (((m[:-1] << 7) | (m[:-1] >> 25)) & 0xFFFFFFFF) ^ m[1:]
    grams = np.unique(np.stack([script[:-1], script[1:]], axis=1), axis=0)
    gm = (grams.astype(np.uint64) * 0x9E3779) & 0xFFFFFFFF
    gx = (((gm[:, 0] << 7) | (gm[:, 0] >> 25)) & 0xFFFFFFFF) ^ gm[:, 1]
    assert len(np.unique(gx)) < len(grams), "no colliding 2-gram hashes: pick another seed"
    # fan works: random tokens with script spans planted, among them the colliding 2-grams
    _, first, counts = np.unique(gx, return_index=True, return_counts=True)
    dup_hash = gx[first[counts > 1]]
    dup = grams[np.isin(gx, dup_hash)]
    works = []
    for w in range(12):
        t = rng.integers(0, len(words), size=400).astype(np.uint32)
        for _ in range(6):
            a = int(rng.integers(0, len(script) - 12))
            b = int(rng.integers(0, len(t) - 12))
            t[b:b + 8] = script[a:a + 8]
        for j, g in enumerate(dup[:40]):
            t[10 + 9 * j:12 + 9 * j] = g
        works.append(t)
    off = np.zeros(len(works) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(t) for t in works])
    tok = np.concatenate(works)
    cfg = abi.make_config(window_size=2)
    from fandom_search_amd.engine import ScriptIndex
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(2), cfg=cfg)
    assert ix.info["path"] == abi.FS_MODE_EXACT
    got, st = ix.search(ix.corpus(tok, off, synth_base["chars"], synth_base["off"]))
    oi = util.oracle_index(cfg, script, words, emb, synth.lsh_normals(2), threads=8)
    want, ost = oi.search(tok, off, synth_base["chars"], synth_base["off"])
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches and len(got) > 100
    ix.close()


def test_direct_and_bitmap_paths_agree(synth_base, monkeypatch):
    """Up to 256 MiB of ids the scan writes candidate records per wave range and
    k_verify_direct reads them (no bitmap, no expand kernel); FS_SCAN_DIRECT=0
    keeps the bitmap + k_expand path.  Same rows, same statistics, also when the
    record lists start far too short (FS_SCAN_CAPW=2) and are grown."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(5000)
    # a work that quotes a long stretch of the script: every lane of several sub-tiles
    # holds candidates, the worst case for the record lists
    tok, off = util.ragged_corpus([1500] * 40 + [0, 5, 6, 3000], script)
    tok = tok.copy()
    tok[int(off[3]):int(off[3]) + 1400] = script[100:1500]
    results = []
    monkeypatch.setenv("FS_SCAN_ROWS", "0")       # the chained kernels (k_scan_rows: below)
    for env in ({}, {"FS_SCAN_DIRECT": "0"}, {"FS_SCAN_CAPW": "2"}):
        for k in ("FS_SCAN_DIRECT", "FS_SCAN_CAPW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                         cfg=abi.make_config())
        c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
        rows, st = ix.search(c)
        rows2, st2 = ix.search(c)                 # second search starts with the grown lists
        assert rows.tobytes() == rows2.tobytes()
        results.append((rows.tobytes(), st.candidates, st.matches, st.rows))
        ix.close()
    assert results[0] == results[1] == results[2]
    oi = util.oracle_index(abi.make_config(), script, words, emb, synth.lsh_normals(6))
    want, ost = oi.search(tok, off, synth_base["chars"], synth_base["off"])
    assert results[0][0] == want.tobytes() and results[0][2] == ost.matches


def test_eight_byte_wire_rows_round_trip(synth_base):
    """8-byte wire records {token position, orig_ix | k << 18 | lev << 22} expand,
    with the batch's work offsets, to the same fs_row bytes (ragged and empty works
    included)."""
    import torch
    from fandom_search_amd import _lib
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(6000)
    tok, off = util.ragged_corpus([1200] * 40 + [0, 5, 0, 0, 3000, 7], script)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    corpus = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    want, _ = ix.search(corpus)
    cap = len(want) + 10
    packed = torch.zeros(cap * 8, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # (as below: the fill is torch's, the search writes on the library's streams)
    n, st = ix.search_device(corpus, packed.data_ptr(), cap, packed=8)
    assert n == len(want)
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    full = torch.empty(cap * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # torch's fill / copy is complete before the library's own streams touch the buffer (engine.torch_ready)
    ix.unpack8_device(packed.data_ptr(), n, d_off.data_ptr(), len(off) - 1, full.data_ptr())
    got = full.cpu().numpy()[:n * 32].view(abi.ROW_DTYPE)
    assert got.tobytes() == want.tobytes()
    with pytest.raises(_lib.FsError) as e:
        ix.search_device(corpus, packed.data_ptr(), 10, packed=8)
    assert e.value.code == abi.FS_E_CAPACITY and e.value.required == len(want)


def test_rows_header_receives_the_count(synth_base):
    """FS_ROWS_HEADER: the record count lands in the first eight bytes of a 32-byte
    header in front of the records (all three device record formats)."""
    import torch
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(4000)
    tok, off = util.ragged_corpus([900] * 25 + [0, 4], script)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    corpus = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    want, _ = ix.search(corpus)
    cap = len(want) + 5
    for packed, size in ((False, 32), (True, 16), (8, 8)):
        buf = torch.full((32 + cap * size,), 0xAB, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()      # torch's fill / copy is complete before the library's own streams touch the buffer (engine.torch_ready)
        n, st = ix.search_end(ix.search_begin(corpus, buf.data_ptr(), cap, packed=packed, header=True))
        host = buf.cpu().numpy()
        assert n == len(want) and int(host[:8].view(np.uint64)[0]) == n
        if not packed:
            assert host[32:32 + n * 32].tobytes() == want.tobytes()
    ix.close()


def test_corpus_may_outlive_its_index(synth_base):
    """Destroying an index before its corpora must be harmless: through the Python
    wrapper (closes the corpora first) and at the C ABI (the index detaches them, a
    detached corpus can only be destroyed)."""
    import ctypes as C
    from fandom_search_amd import _lib
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(2000)
    tok, off = util.ragged_corpus([500] * 6, script)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    rows, _ = ix.search(c)
    ix.close()                                   # closes c as well
    assert not c._h
    # C ABI order: index first, then the corpus
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    L = _lib.load()
    ix._corpora.discard(c)
    L.fs_index_destroy(ix._h)
    ix._h = C.c_void_p()
    assert L.fs_corpus_update_end(c._h) == abi.FS_OK          # nothing pending: no-op
    rc = L.fs_corpus_update_begin(c._h, abi.ptr(c.tok_vec, C.c_uint32), None,
                                  abi.ptr(c.work_off, C.c_uint64), c.n_works)
    assert rc == abi.FS_E_INVALID
    c.close()


def test_dense_chunks_span_several_verify_tiles(synth_base, monkeypatch):
    """3.3 M tokens: a chunk of the direct path is three sub-tiles (1536 windows).
    Works that quote 1900 script tokens in a row fill whole chunks with candidates,
    so k_verify_direct stages more than one 1024-candidate tile per chunk and the
    record lists grow; the bitmap form of the pipeline must give the same bytes, and
    the quoted works alone must equal the oracle."""
    from fandom_search_amd.engine import ScriptIndex
    words, emb = synth_base["words"], synth_base["emb"]
    script = synth.script_tokens(20000)
    tok, off = synth.corpus_tokens(1650, 2000, script)
    tok = tok.copy()
    dense = [3, 400, 401, 1649]
    for j, w in enumerate(dense):
        a = int(off[w]) + 17 * j
        tok[a:a + 1900] = script[500 + 1000 * j:2400 + 1000 * j]
    swords = [words[int(t)] for t in script]
    ix = ScriptIndex(script, swords, emb, synth.lsh_normals(6), cfg=abi.make_config())
    c = ix.corpus(tok, off, synth_base["chars"], synth_base["off"])
    rows, st = ix.search(c)
    monkeypatch.setenv("FS_SCAN_DIRECT", "0")
    ix.reload_switches()
    brows, bst = ix.search(c)
    monkeypatch.delenv("FS_SCAN_DIRECT")
    ix.reload_switches()
    assert rows.tobytes() == brows.tobytes()
    # (the candidate counts differ: k_scan_rows tests runs of script 4-grams, the chained
    # kernels the whole 6-gram)
    assert st.matches == bst.matches and st.rows == bst.rows
    ix.close()
    # the dense works on their own against the oracle
    sub_tok = np.concatenate([tok[int(off[w]):int(off[w + 1])] for w in dense])
    sub_off = np.arange(len(dense) + 1, dtype=np.uint64) * np.uint64(2000)
    oi = util.oracle_index(abi.make_config(), script, words, emb, synth.lsh_normals(6))
    want, _ = oi.search(sub_tok, sub_off, synth_base["chars"], synth_base["off"])
    got = rows[np.isin(rows["work"], dense)].copy()
    got["work"] = np.searchsorted(dense, got["work"])
    util.assert_rows_equal(got, want)
    assert len(want) > 4 * 1800


@pytest.mark.parametrize("n", [2, 4, 5, 6, 7, 8])
def test_scan_rows_and_chain_agree(synth_base, monkeypatch, n):
    """k_scan_rows (tokens -> records in one kernel, the default) against the chain it
    replaces (FS_SCAN_ROWS=0: k_scan8, k_verify_direct, k_hitrows, k_rows) and the oracle:
    same bytes and statistics in all three record formats, also when the staging area
    starts far too small (FS_RANGES_CAPROW=2), with both ways of putting the records into
    place (inside the launch, or k_compact: FS_ROWS_FINISH=2), when the in-launch wait
    gives up at once (FS_WAIT_SPINS=0: the search is flagged and repeated through the
    chained kernels), with the Bloom test of the whole n-gram in place of the runs of
    script K-grams (FS_SCAN_SUB=0), with the launch shape of an index that overlaps searches
    (FS_LANES=4: two workgroups of eight wave ranges per CU), with a sub-shingle filter far
    too small (many false candidates), with the displacement seeds read from memory instead
    of LDS (as for scripts with more than 16 K buckets), with the shared rounds of round 4 switched off or starved
    of pool space (the dense work below posts slices from every wave range it covers), with works that quote long stretches of the script (several rounds
    of candidates per flush, hits carried from round to round), hits at range and work
    boundaries, ragged and empty works.  n = 7, 8 need a table the exact-n-gram proof
    accepts: 256 one-hot vectors."""
    import torch
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings
    if n <= 6:
        words, emb, V = synth_base["words"], synth_base["emb"], synth.VOCAB_SIZE
        chars, coff = synth_base["chars"], synth_base["off"]
    else:
        V = 256
        words = synth.vocab_words(V)
        emb = np.eye(V, synth.EMB_DIM, dtype=np.float32)
        chars, coff = pack_strings(words)
    script = synth.script_tokens(5000, vocab_size=V)
    lengths = [1500] * 40 + [0, 5, n, n - 1, 3000, 511, 512, 513]
    parts = [synth.fanwork_tokens(i, L, script, V) if L else np.zeros(0, np.uint32)
             for i, L in enumerate(lengths)]
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    tok = np.concatenate(parts).astype(np.uint32)
    tok[int(off[3]):int(off[3]) + 1400] = script[100:1500]      # dense: every window a hit
    tok[int(off[7]) + 1490:int(off[7]) + 1500] = script[40:50]  # hit at the end of a work ...
    tok[int(off[8]):int(off[8]) + 10] = script[40:50]           # ... and at the start of the next
    for k in range(9, 30):                                      # hits straddling 512-token borders
        at = (int(off[k]) // 512 + 1) * 512 - (k % (n + 3))
        tok[at:at + n + 2] = script[700 + k:700 + k + n + 2]
    # quotes of 90, 150 and 300 tokens: more candidates than one round takes, few enough
    # slices for the shared rounds (one, two, five per wave range); one across a 512-token border
    for k, (ln, where) in enumerate(((90, 200), (150, 300), (300, 100), (150, 450))):
        at = (int(off[31 + k]) // 512 + 1) * 512 + where
        tok[at:at + ln] = script[2000 + 400 * k:2000 + 400 * k + ln]
    normals = synth.lsh_normals(n)
    cfg = abi.make_config(window_size=n)
    results = []
    envs = ({}, {"FS_SCAN_ROWS": "0"}, {"FS_RANGES_CAPROW": "2"}, {"FS_WAIT_SPINS": "0"},
            {"FS_ROWS_FINISH": "2"}, {"FS_ROWS_FINISH": "2", "FS_RANGES_CAPROW": "2"},
            {"FS_SCAN_SUB": "0"}, {"FS_LANES": "4"}, {"FS_LANES": "4", "FS_RANGES_CAPROW": "2"},
            {"FS_SFILTER_LOG2_WORDS": "10"}, {"FS_ROWS_DISP_LDS": "0"},
            # round 4: without the shared rounds (every wave works its own queue off), with a
            # pool for the slices' records that is far too small (the search reports what it
            # needs and is repeated), with the pool AND the staging area too small, with all
            # records through the staging area (none kept in registers), without the scanners'
            # priority, with equal and with lopsided shares of the waves of a SIMD, with the
            # phase clocks of the timeline running (FS_DIAG 6)
            {"FS_ROWS_COOP": "0"}, {"FS_ROWS_XPOOL": "8"}, {"FS_ROWS_XPOOL": "8", "FS_RANGES_CAPROW": "2"},
            {"FS_DIAG": "32"}, {"FS_DIAG": "128"}, {"FS_DIAG": "16"}, {"FS_ROWS_SHARES": "400,300,200,124"},
            {"FS_DIAG": "6"}, {"FS_ROWS_FINISH": "1", "FS_LANES": "2"})
    for env in envs:
        for k in ("FS_SCAN_ROWS", "FS_RANGES_CAPROW", "FS_WAIT_SPINS", "FS_ROWS_FINISH", "FS_SCAN_SUB",
                  "FS_LANES", "FS_SFILTER_LOG2_WORDS", "FS_ROWS_DISP_LDS", "FS_ROWS_COOP", "FS_ROWS_XPOOL",
                  "FS_DIAG", "FS_ROWS_SHARES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ix = ScriptIndex(script, [words[int(t)] for t in script], emb, normals, cfg=cfg)
        assert ix.info["path"] == abi.FS_MODE_EXACT
        c = ix.corpus(tok, off, chars, coff)
        rows, st = ix.search(c)
        rows2, _ = ix.search(c)
        assert rows.tobytes() == rows2.tobytes()
        # the hand-off's give-up is counted per call (and nowhere else does one happen)
        if n <= 8 and "FS_SCAN_ROWS" not in env:
            fell_back = env.get("FS_WAIT_SPINS") == "0" and "FS_LANES" not in env
            assert (st.handoff_fallbacks > 0) == fell_back, (env, st.handoff_fallbacks)
        wires = []
        cap = len(rows) + 3
        for packed, size in ((True, 16), (8, 8)):
            buf = torch.zeros(32 + cap * size, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()      # torch's fill / copy is complete before the library's own streams touch the buffer (engine.torch_ready)
            nw, _ = ix.search_end(ix.search_begin(c, buf.data_ptr(), cap, packed=packed, header=True))
            host = buf.cpu().numpy()
            assert nw == len(rows) and int(host[:8].view(np.uint64)[0]) == nw
            wires.append(host[32:32 + nw * size].tobytes())
        # (st.candidates depends on the filter: K-gram runs, or the Bloom test of the n-gram)
        results.append((rows.tobytes(), st.matches, st.rows, wires))
        ix.close()
    assert all(r == results[0] for r in results[1:])
    oi = util.oracle_index(cfg, script, words, emb, normals)
    want, ost = oi.search(tok, off, chars, coff)
    assert results[0][0] == want.tobytes() and results[0][1] == ost.matches
    assert len(want) > 1400
    oi.close()
