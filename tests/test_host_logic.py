"""Host-side logic that needs no GPU: work listing, CSV bytes, result naming,
script markup parsing and linting, vocabulary ids, the matrix command."""

import csv
import datetime
import os
import types

import numpy as np
import pytest

from fandom_search_amd import abi, matrix, search, synth, vocab
from fandom_search_amd.cli import build_parser


def test_list_fan_works_is_sorted_then_seeded_shuffle(tmp_path):
    for i in range(10):
        (tmp_path / ("w%d.txt" % i)).write_text("x")
    got = [os.path.basename(p) for p in search.list_fan_works(str(tmp_path))]
    # random.seed(4815162342); shuffle(range(10)) == [5,4,2,3,1,8,7,0,6,9]
    assert got == ["w%d.txt" % i for i in [5, 4, 2, 3, 1, 8, 7, 0, 6, 9]]
    assert [os.path.basename(p) for p in search.list_fan_works(str(tmp_path), 2, 3)] == \
        ["w2.txt", "w3.txt", "w1.txt"]
    assert len(search.list_fan_works(str(tmp_path), -5, -1)) == 10


def test_listing_order_os_shuffles_the_listing_as_it_comes(tmp_path, monkeypatch):
    """--listing os / FANDOM_SEARCH_LISTING=os: os.listdir()'s order goes into the seeded
    shuffle untouched, as in the reference (search.py:349,354-355); the default sorts first."""
    import random
    names = ["w%d.txt" % i for i in (7, 2, 9, 0, 4, 1, 8, 3, 6, 5)]
    for n in names:
        (tmp_path / n).write_text("x")
    real = os.listdir
    monkeypatch.setattr(os, "listdir", lambda d: list(names) if str(d) == str(tmp_path) else real(d))
    want = list(names)
    random.seed(search.SHUFFLE_SEED)
    random.shuffle(want)
    monkeypatch.setenv("FANDOM_SEARCH_LISTING", "os")
    got = [os.path.basename(p) for p in search.list_fan_works(str(tmp_path))]
    assert got == want
    assert [os.path.basename(p) for p in search.list_fan_works(str(tmp_path), 2, 3)] == want[2:5]
    monkeypatch.delenv("FANDOM_SEARCH_LISTING")
    assert [os.path.basename(p) for p in search.list_fan_works(str(tmp_path))] == \
        ["w%d.txt" % i for i in [5, 4, 2, 3, 1, 8, 7, 0, 6, 9]]
    assert [os.path.basename(p) for p in search.list_fan_works(str(tmp_path), order="os")] == want
    monkeypatch.setenv("FANDOM_SEARCH_LISTING", "alphabetical")
    with pytest.raises(ValueError):
        search.list_fan_works(str(tmp_path))
    args = build_parser().parse_args(["search", "fan", "script", "--listing", "os"])
    assert args.listing == "os"


def test_threads_and_workers_follow_the_cpu_share(monkeypatch):
    """usable_cpus(): the affinity mask cut down to the control group's quota; the encoder's
    threads and the forked workers share it where both run, and a one-rank run whose text and
    batch files are both native forks nobody."""
    for k in ("FANDOM_SEARCH_WORKERS", "FANDOM_SEARCH_TEXT_THREADS", "FANDOM_SEARCH_NATIVE_CSV",
              "FANDOM_SEARCH_NATIVE_TEXT", "FANDOM_SEARCH_TOKENIZER", "FANDOM_SEARCH_TEXT_HANDLES"):
        monkeypatch.delenv(k, raising=False)
    cpus = search.usable_cpus()
    assert 1 <= cpus <= len(os.sched_getaffinity(0))
    assert search.default_workers(1) == 0                      # native text + native csv, one rank
    assert search.default_text_threads(1) == max(1, min(16, cpus // 2))         # two encoders share them
    assert search.default_workers(2) == max(1, min(16, (cpus // 2 + 1) // 2))     # ranks write through the pool
    monkeypatch.setenv("FANDOM_SEARCH_NATIVE_CSV", "0")
    assert search.default_workers(1) == max(1, min(16, (cpus + 1) // 2))
    assert search.default_text_threads(1) == max(1, min(16, cpus // 2 // 2))
    monkeypatch.setenv("FANDOM_SEARCH_TEXT_HANDLES", "1")
    assert search.default_text_threads(1) == max(1, min(16, cpus // 2))
    monkeypatch.setenv("FANDOM_SEARCH_NATIVE_TEXT", "0")
    assert search.default_workers(1) == max(1, min(16, cpus))
    monkeypatch.setenv("FANDOM_SEARCH_WORKERS", "3")
    assert search.default_workers(1) == 3 and search.default_text_threads(1) == 3
    monkeypatch.setenv("FANDOM_SEARCH_TEXT_THREADS", "5")
    assert search.default_text_threads(1) == 5


def test_write_records_bytes(tmp_path):
    """csv.writer defaults: \\r\\n terminators, minimal quoting, None -> empty,
    floats by repr (search.py:331-334)."""
    p = tmp_path / "r.csv"
    search.write_records([["a.txt", 3, 'say "hi", ok', 12345678901234567890, 7, "w", 1, None, 2,
                           1.1102230246251565e-16, 7, 7.771561172376096e-16]], str(p))
    assert p.read_bytes() == (b'a.txt,3,"say ""hi"", ok",12345678901234567890,7,w,1,,2,'
                              b'1.1102230246251565e-16,7,7.771561172376096e-16\r\n')


def test_unused_result_name(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    today = '{:%Y%m%d}'.format(datetime.date.today())
    base = 'match-6gram{}'
    assert search.unused_result_name(base) == 'match-6gram-%s.csv' % today
    open('match-6gram-%s.csv' % today, 'w').close()
    assert search.unused_result_name(base) == 'match-6gram-%s-1.csv' % today
    open('match-6gram-%s-1.csv' % today, 'w').close()
    assert search.unused_result_name(base) == 'match-6gram-%s-2.csv' % today


def test_load_markup_script_rows(tmp_path):
    p = tmp_path / "s.txt"
    p.write_text("SCENE_NUMBER<<12A>>\nCHARACTER_NAME<<REY>>\nLINE<<Hello there, General>>\n"
                 "DIRECTION<<She leaves>>\nSCENE_NUMBER<<INT>>\nLINE<<Run>>\n"
                 "SCENE_NUMBER<<40>>\nCHARACTER_NAME<<FINN>>\nLINE<<go  now>>\n")
    rows = search.load_markup_script(str(p))
    assert rows[0] == ['LOWERCASE', 'SPACY_ORTH_ID', 'SCENE', 'CHARACTER']
    body = rows[1:]
    assert [r[0] for r in body] == ["hello", "there", ",", "general", "run", "go", "now"]
    assert body[0][1] == vocab.hash_string("hello")
    # '12A' -> 12; 'INT' has no digits -> from then on the running tag count
    assert [r[2] for r in body] == [12, 12, 12, 12, 2, 3, 3]
    assert [r[3] for r in body] == ["REY"] * 5 + ["FINN"] * 2


def test_scene_label_that_isdigit_accepts_and_int_refuses(tmp_path):
    """'\u00b2' passes str.isdigit() but int() raises ValueError: the reference catches
    it and switches to the running scene count for the rest of the file
    (search.py:309-317)."""
    p = tmp_path / "s.txt"
    p.write_text("SCENE_NUMBER<<7>>\nLINE<<one>>\nSCENE_NUMBER<<\u00b2>>\nLINE<<two>>\n"
                 "SCENE_NUMBER<<90>>\nLINE<<three>>\n", encoding="utf-8")
    body = search.load_markup_script(str(p))[1:]
    assert [r[2] for r in body] == [7, 2, 3]


def test_vocabulary_must_be_named(monkeypatch):
    """No silent fall-back to the synthetic vocabulary (random vectors)."""
    monkeypatch.delenv("FANDOM_SEARCH_VECTORS", raising=False)
    monkeypatch.delenv("FANDOM_SEARCH_SYNTHETIC_VOCAB", raising=False)
    search.set_vocab(None)
    with pytest.raises(RuntimeError, match="no vector table"):
        search.get_vocab()
    monkeypatch.setenv("FANDOM_SEARCH_SYNTHETIC_VOCAB", "1")
    try:
        assert search.get_vocab().dim == synth.EMB_DIM
    finally:
        search.set_vocab(None)


def test_synthetic_script_round_trip(tmp_path):
    words = synth.vocab_words()
    script = synth.script_tokens(1234)
    p = tmp_path / "script.txt"
    p.write_text(synth.script_markup(script, words))
    rows = search.load_markup_script(str(p))[1:]
    scene, char = synth.script_columns(len(script))
    assert [r[0] for r in rows] == [words[t] for t in script]
    assert [r[2] for r in rows] == scene.tolist() and [r[3] for r in rows] == char
    assert search.validate_markup_script(str(p)) is True


def test_validate_reports_errors(tmp_path, capsys):
    p = tmp_path / "bad.txt"
    p.write_text("LINE<<ok>>\nLINE<<unbalanced << left>>\nLYNE<<typo>>\nLINE<<x>> >>\n")
    assert search.validate_markup_script(str(p)) is False
    out = capsys.readouterr().out
    assert "Unbalanced left tag delimiters:" in out and "On line 2" in out
    assert "Unbalanced right tag delimiters:" in out
    assert "Unexpected tag labels:" in out and "LYNE" in out


def test_vocab_ids_and_oov_encoding():
    words = ["alpha", "beta"]
    emb = np.eye(2, 8, dtype=np.float32)
    v = vocab.Vocab(words, emb, oov_hash=lambda s: {"zz": 3, "zzzz": 1, "zzzzzz": 3}.get(s, 0))
    sids, vids = v.encode(["beta", "zz", "alpha", "zz"])
    assert vids[0] == 1 and vids[2] == 0 and sids[1] == sids[3] == 2
    assert vids[1] & abi.FS_OOV_FLAG
    code = int(vids[1]) & 0x7FFFFFFF
    assert (code // 64, (code // 8) % 8, code % 8) == (1, 3, 3)      # sorted hot positions
    assert v.vector(2).tolist() == [0, 1, 0, 1, 0, 0, 0, 0]
    chars, off = v.string_table()
    assert off.tolist() == [0, 5, 9, 11] and chars.dtype == np.uint32


def test_chunk_text_cuts_at_spaces():
    txt = ("ab " * 40000).strip()                      # 119999 chars
    parts = list(vocab.chunk_text(txt))
    assert len(parts) == 2 and all(len(p) <= 100000 for p in parts)
    assert " ".join(parts) == txt
    assert list(vocab.chunk_text("short text")) == ["short text"]


def test_cli_shape():
    ap = build_parser()
    a = ap.parse_args(["search", "fandir", "script.txt", "-n", "5", "-s", "2"])
    assert (a.fan_works, a.script, a.num_works, a.skip_works) == ("fandir", "script.txt", 5, 2)
    a = ap.parse_args(["search", "d", "s"])
    assert (a.num_works, a.skip_works) == (-1, 0)
    a = ap.parse_args(["matrix", "in.csv", "sw", "-n", "4"])
    assert (a.i, a.m, a.n) == ("in.csv", "sw", 4)
    assert ap.parse_args(["matrix", "in.csv", "sw"]).n == 6


def _match_csv(path, spans):
    """spans: (work, fan_start, script_start, length)."""
    script_words = ["W%d" % i for i in range(200)]
    with open(path, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(search.new_record_structure['fields'])
        for work, f0, s0, ln in spans:
            for k in range(ln):
                wr.writerow([work, f0 + k, "x", 1, s0 + k, script_words[s0 + k], 1, "C", 1,
                             0.0, 7, 0.0])


def test_matrix_intent(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    # three works quote script words 10..17 (three 6-gram starts 10,11,12);
    # start 11 is the most common start overall, so every span is represented
    # by the n-gram starting at 11; a 5-word span is too short to count
    _match_csv("m.csv", [("a.txt", 0, 10, 8), ("b.txt", 5, 11, 6), ("c.txt", 9, 11, 7),
                         ("c.txt", 40, 50, 5), ("d.txt", 0, 100, 6)])
    out = matrix.process(types.SimpleNamespace(i="m.csv", m="sw", n=6))
    assert out == "sw-most-common-perfect-matches-no-overlap-6-gram-match-matrix.csv"
    rows = list(csv.reader(open(out)))
    assert rows[0] == ["FILENAME", "w11 w12 w13 w14 w15 w16", "w100 w101 w102 w103 w104 w105"]
    assert rows[1] == ["(total)", "3", "1"]
    assert rows[2:] == [["a.txt", "1", "0"], ["b.txt", "1", "0"], ["c.txt", "1", "0"],
                        ["d.txt", "0", "1"]]


def test_matrix_empty(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    _match_csv("m.csv", [("a.txt", 0, 10, 3)])
    out = matrix.process(types.SimpleNamespace(i="m.csv", m="x", n=6))
    assert list(csv.reader(open(out))) == [["FILENAME"], ["(total)"]]


def test_vector_table_with_shared_and_zero_rows(tmp_path, monkeypatch):
    """The .npz of tools/export_spacy_vectors.py: words that share a row (spaCy's key2row; most
    keys of en_core_web_md do) share a vector id, a word on an all-zero row keeps its row
    (has_vector is true for it in the reference, search.py:74-75), a word without a row is OOV."""
    import numpy as np
    from fandom_search_amd import search, vocab as vocab_mod
    vec = np.zeros((4, 300), dtype=np.float32)
    vec[0, 0] = 1.0
    vec[1, 1] = 1.0
    vec[2, 2] = 1.0                       # row 3 stays all zero
    path = tmp_path / "vectors.npz"
    np.savez(path, words=np.array(["cat", "feline", "dog", "nil", "kitty"]),
             rows=np.array([0, 0, 1, 3, 0]), vectors=vec)
    monkeypatch.setenv("FANDOM_SEARCH_VECTORS", str(path))
    monkeypatch.setattr(search, "_VOCAB", None)
    v = search.get_vocab()
    sid, vid = v.encode(["cat", "feline", "kitty", "dog", "nil", "unknown", "cat"])
    assert vid[0] == vid[1] == vid[2] == 0 and vid[3] == 1 and vid[4] == 3 and vid[6] == 0
    assert len(set(sid[:5].tolist())) == 5            # five different strings
    assert vid[5] & vocab_mod.OOV_FLAG and v.has_vector(int(sid[4])) and not v.has_vector(int(sid[5]))
    assert not np.any(v.vector(int(sid[4])))
    with pytest.raises(ValueError):
        vocab_mod.Vocab(["a", "b"], vec, rows=[0, 9])


def test_batch_csv_from_a_pool_worker_equals_the_inline_form(tmp_path, monkeypatch):
    """ao3.py search hands every batch CSV to a forked worker (search._write_batch: the worker
    parses the script's columns itself and gets the numeric rows as bytes); the file must be
    what join_records + write_records give in the parent, also from a real worker process."""
    import multiprocessing
    monkeypatch.setenv("FANDOM_SEARCH_SYNTHETIC_VOCAB", "1")
    words = synth.vocab_words()
    script = synth.script_tokens(200)
    spath = tmp_path / "script.txt"
    spath.write_text(synth.script_markup(script, words), encoding="utf8")
    orig = search.load_markup_script(str(spath))[1:]
    rng = np.random.default_rng(3)
    n = 300
    rows = np.zeros(n, dtype=abi.ROW_DTYPE)
    rows["work"] = np.sort(rng.integers(0, 4, n))
    rows["fan_ix"] = np.arange(n)
    rows["orig_ix"] = rng.integers(0, len(orig), n)
    rows["lev"] = rng.integers(0, 30, n)
    rows["dist"] = rng.random(n) * 1e-15 - 2e-16
    rows["comb"] = rows["dist"] * rows["lev"]
    filenames = ["fan/%d.txt" % i for i in range(4)]
    fan_words = [words[int(i)] if i % 7 else "Café," for i in rng.integers(0, len(words), n)]
    inline = tmp_path / "inline.csv"
    search.write_records(search.join_records(
        filenames, rows, fan_words, tuple(r[0] for r in orig), tuple(r[1] for r in orig),
        tuple(r[3] for r in orig), tuple(r[2] for r in orig)), str(inline))
    direct = tmp_path / "direct.csv"
    assert search._write_batch(str(spath), str(direct), filenames, rows.tobytes(), fan_words) == n
    with multiprocessing.get_context("fork").Pool(1) as pool:
        forked = tmp_path / "forked.csv"
        assert pool.apply(search._write_batch, (str(spath), str(forked), filenames, rows.tobytes(), fan_words)) == n
    assert inline.read_bytes() == direct.read_bytes() == forked.read_bytes() and inline.stat().st_size > 0


def test_token_pool_survives_a_lost_worker_and_drops_stale_jobs(tmp_path):
    """a tokeniser process that dies must not be replaced by a fork of the (by then
    GPU-holding) parent: the pool is shut down and the text work done inline; jobs queued for a
    list nobody asks for are dropped.  The results are the same either way."""
    import os
    import signal
    import time
    from fandom_search_amd import search
    files = []
    for i in range(6):
        p = tmp_path / ("w%d.txt" % i)
        p.write_text(" ".join("tok%d" % ((i * 7 + j) % 11) for j in range(40 + i)))
        files.append(str(p))
    want = [search.tokenize_files(files[:3])]
    pool = search.TokenPool(2)
    try:
        pool.start(files[:3])
        got = pool.get(files[:3])
        assert len(got) == 1 or sum(len(g[0]) for g in got) == 3
        assert np.concatenate([g[1] for g in got]).tolist() == want[0][1].tolist()
        # stale jobs: three lists queued, the first never asked for
        pool.start(files[3:4]); pool.start(files[4:5]); pool.start(files[5:6])
        pool.get(files[5:6])
        assert len(pool.pending) <= 2
        # a worker dies: the next get() notices, closes the pool and works inline
        os.kill(pool.pool._pool[0].pid, signal.SIGKILL)
        time.sleep(0.3)
        pool.start(files[:3])
        got = pool.get(files[:3])
        assert pool.pool is None
        assert np.concatenate([g[1] for g in got]).tolist() == want[0][1].tolist()
        assert pool.get(files[3:5])[0][0].tolist() == search.tokenize_files(files[3:5])[0].tolist()
    finally:
        pool.close()


def _slow_write(path, text, delay, parent=0):
    import time
    with open(path, "w", newline="") as f:
        f.write(text[:3])                      # a partial file, should the writer die here
        f.flush()
        if os.getpid() != parent:              # (slow in a worker only)
            time.sleep(delay)
        f.write(text[3:])
    return len(text)


def test_queued_batch_writes_survive_a_lost_worker(tmp_path):
    """a pool that is given up never completes the writes queued on it; waiting for
    them without a limit hung the command (and, with more ranks, everybody else in the next
    collective).  finish_writes() polls, notices the loss and redoes the writes inline; a file
    a killed writer left half written is written again."""
    import os
    import signal
    import time
    from fandom_search_amd import search
    pool = search.TokenPool(2)
    try:
        done = tmp_path / "done.csv"
        pool.write_async(_slow_write, (str(done), "finished\r\n", 0.0))
        time.sleep(0.3)
        a, b = tmp_path / "a.csv", tmp_path / "b.csv"
        pool.write_async(_slow_write, (str(a), "first batch\r\n", 30.0, os.getpid()))
        pool.write_async(_slow_write, (str(b), "second batch\r\n", 30.0, os.getpid()))
        time.sleep(0.3)
        for p in pool.pool._pool:
            os.kill(p.pid, signal.SIGKILL)
        t0 = time.time()
        pool.finish_writes()
        assert time.time() - t0 < 10 and pool.pool is None and not pool.writes
        assert done.read_bytes() == b"finished\r\n"
        assert a.read_bytes() == b"first batch\r\n" and b.read_bytes() == b"second batch\r\n"
        # with the pool gone a write is done at finish_writes() at the latest
        c = tmp_path / "c.csv"
        pool.write_async(_slow_write, (str(c), "third\r\n", 0.0))
        pool.finish_writes()
        assert c.read_bytes() == b"third\r\n"
    finally:
        pool.close()


def test_a_failing_batch_write_is_raised_by_finish_writes(tmp_path):
    from fandom_search_amd import search
    pool = search.TokenPool(2)
    try:
        pool.write_async(_slow_write, (str(tmp_path / "no" / "such" / "dir.csv"), "x", 0.0))
        with pytest.raises(FileNotFoundError):
            pool.finish_writes()
    finally:
        pool.close()


def test_prefilter_rule_for_eight_windows_at_once():
    """k_scan_near8 (fs_scan.hip: window_flags_near8) decides a lane's eight windows with runs of
    set bits instead of a loop over the windows: with T = n - 2 three-gram tests per window and
    R = T - 3 that must hold, window j passes iff for some a in 0..R its first a and its last
    R - a tests hold.  The same answer as the rule as stated -- every failed test within three
    consecutive positions -- for random test outcomes at every supported window size."""
    import random
    K = 3
    rng = random.Random(7)

    def stated(z):
        if z == 0:
            return True
        hi = z.bit_length() - 1
        lo = (z & -z).bit_length() - 1
        return hi - lo < K

    for n in (7, 8, 9, 10, 12):
        T, NB = n - K + 1, 8 + n - K
        R = T - K
        for _ in range(4000):
            bits = rng.getrandbits(NB)
            run = [0xFFFFFFFF, bits]
            for r in range(2, R + 1):
                run.append(run[-1] & (bits >> (r - 1)))
            flags = 0
            for a in range(R + 1):
                flags |= run[a] & (run[R - a] >> (a + K))
            flags &= 0xFF
            for j in range(8):
                z = ~(bits >> j) & ((1 << T) - 1)
                assert stated(z) == bool(flags >> j & 1), (n, bits, j)
