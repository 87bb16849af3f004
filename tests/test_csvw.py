"""The native batch-file formatter (fs_csvw_*, fandom_search_amd/csvw.py) against the Python
path it stands in for -- search.join_records + search.write_records, i.e. the reference's
record fields (search.py:192-218) through csv.writer (search.py:331-334): the same bytes, for
ordinary records, for fields that need quoting (commas, quotes, CR, LF, non-ASCII), for None
columns, for scene numbers beyond 64 bits, and for floats of every magnitude (repr: shortest
digits, positional between 1e-4 and 1e16, else exponent form; -0.0, nan, inf).  No GPU involved."""

import io
import os
import struct

import numpy as np
import pytest

from fandom_search_amd import abi, csvw, search, vocab


def _python_bytes(tmp_path, filenames, rows, words, cols):
    recs = search.join_records(filenames, rows, words, *cols)
    p = tmp_path / "py.csv"
    search.write_records(recs, str(p))
    return p.read_bytes()


def _rows(work, fan_ix, orig_ix, lev, dist, comb):
    r = np.zeros(len(work), dtype=abi.ROW_DTYPE)
    r['work'], r['fan_ix'], r['orig_ix'], r['lev'], r['dist'], r['comb'] = work, fan_ix, orig_ix, lev, dist, comb
    return r


SCRIPT_WORDS = ["luke", "i", "am", "your", "father", 'say "no"', "a,b", "line\nbreak", "cr\rhere", "naïve", "日本語", ""]


def _script_cols():
    n = len(SCRIPT_WORDS)
    orth = [vocab.hash_string(w) for w in SCRIPT_WORDS]
    character = ["VADER", None, "LUKE, SON", 'THE "EMPEROR"', "", "R2\nD2", None, "HAN", "LEIA", "C-3PO", None, "YODA"][:n]
    scene = [1, 2, None, 10 ** 25, 0, 7, 7, None, 12, 13, 14, 15][:n]
    return SCRIPT_WORDS, orth, character, scene


def test_bytes_equal_csv_writer_on_awkward_fields(tmp_path):
    cols = _script_cols()
    strings = ["Luke", "father", 'quo"te', "com,ma", "new\nline", "ret\rurn", "ünï", "𝔘𝔫𝔦", "", "plain"]
    filenames = [str(tmp_path / "w1.txt"), str(tmp_path / 'dir,with "odd" name.txt'), "rel/ative.txt"]
    n = 60
    rng = np.random.default_rng(1)
    sids = rng.integers(0, len(strings), size=n).astype(np.uint32)
    rows = _rows(rng.integers(0, len(filenames), n), rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32),
                 rng.integers(0, len(SCRIPT_WORDS), n), rng.integers(0, 90, n),
                 rng.random(n) * 0.1, rng.random(n) * 5)
    w = csvw.CsvWriter(*cols, strings=strings)
    got = w.format(filenames, rows, sids)
    want = _python_bytes(tmp_path, filenames, rows, [strings[int(s)] for s in sids], cols)
    assert got == want
    assert b"\r\n" in got and b'"com,ma"' in got and b'"quo""te"' in got
    # the vocabulary grows between batches: the writer follows
    strings += ["later", "añadido"]
    sids2 = np.array([len(strings) - 1, len(strings) - 2, 0], dtype=np.uint32)
    got2 = w.format(filenames, rows[:3], sids2)
    assert got2 == _python_bytes(tmp_path, filenames, rows[:3], [strings[int(s)] for s in sids2], cols)
    # an empty batch is an empty file
    assert w.format(filenames, rows[:0], sids[:0]) == b""
    # what does not fit the tables is refused, not written
    bad = rows[:1].copy()
    bad['orig_ix'] = len(SCRIPT_WORDS)
    with pytest.raises(Exception):
        w.format(filenames, bad, sids[:1])
    with pytest.raises(Exception):
        w.format(filenames, rows[:1], np.array([len(strings)], dtype=np.uint32))


def test_floats_are_pythons_repr(tmp_path):
    cols = _script_cols()
    strings = ["x"]
    special = [0.0, -0.0, 1.0, -1.0, 0.1, 1.1102230246251565e-16, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9999999999999998.0,
               1.5e16, 123456789012345678.0, 5e-324, 1.7976931348623157e308, 2.2250738585072014e-308,
               float("nan"), float("inf"), float("-inf"), 0.30000000000000004, 1 / 3, 2 / 3, 100.0, 1e22, 1e23,
               0.05131670194948623, 7.771561172376096e-16, 123456.789, 0.001, 0.0001234, 12345678901234567890.0]
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2 ** 64, size=150000, dtype=np.uint64)
    anyd = np.frombuffer(bits.tobytes(), dtype=np.float64)              # every exponent, denormals, nans
    small = rng.random(150000) * 10.0 ** rng.integers(-20, 20, size=150000)
    vals = np.concatenate([np.array(special), anyd, small, -small[:1000]])
    n = len(vals)
    rows = _rows(np.zeros(n, np.uint32), np.arange(n, dtype=np.uint32), np.zeros(n, np.uint32),
                 np.ones(n, np.uint32), vals, vals[::-1].copy())
    w = csvw.CsvWriter(*cols, strings=strings)
    got = w.format(["f"], rows, np.zeros(n, np.uint32)).split(b"\r\n")
    assert got[-1] == b"" and len(got) == n + 1
    for i in (list(range(len(special))) + rng.integers(0, n, size=20000).tolist()):
        f = got[i].split(b",")
        assert f[9].decode() == repr(float(vals[i])), (i, vals[i])
        assert f[11].decode() == repr(float(vals[n - 1 - i])), (i, vals[n - 1 - i])
    # ... and all of them, as one digest against the Python writer
    want = _python_bytes(tmp_path, ["f"], rows, ["x"] * n, cols)
    assert b"\r\n".join(got) == want


def test_spacy_keys_of_the_fan_words():
    """FAN_WORK_ORTH_ID is made natively (MurmurHash64A, seed 1): the four hashes spaCy's
    documentation prints, and vocab.hash_string on strings of every length mod 8."""
    cols = _script_cols()
    strings = ["coffee", "apple", "", "a", "ab", "abcdefg", "abcdefgh", "abcdefghi", "ünïcödé strïng", "x" * 100]
    w = csvw.CsvWriter(*cols, strings=strings)
    n = len(strings)
    rows = _rows(np.zeros(n, np.uint32), np.arange(n), np.zeros(n, np.uint32), np.ones(n, np.uint32),
                 np.zeros(n), np.zeros(n))
    lines = w.format(["f"], rows, np.arange(n, dtype=np.uint32)).split(b"\r\n")[:-1]
    for s, line in zip(strings, lines):
        assert int(line.split(b",")[3]) == vocab.hash_string(s), s
    assert int(lines[0].split(b",")[3]) == 3197928453018144401      # spaCy: nlp.vocab.strings["coffee"]


def test_write_async_and_switch(tmp_path, monkeypatch):
    cols = _script_cols()
    strings = ["a", "b"]
    w = csvw.CsvWriter(*cols, strings=strings)
    rows = _rows([0, 0], [3, 4], [0, 1], [7, 8], [0.0, 0.5], [0.0, 4.0])
    for k in range(12):
        w.write_async(str(tmp_path / ("b%d.csv" % k)), ["f.txt"], rows, [0, 1])
    w.finish()
    want = _python_bytes(tmp_path, ["f.txt"], rows, ["a", "b"], cols)
    assert all((tmp_path / ("b%d.csv" % k)).read_bytes() == want for k in range(12))
    w.write_async(str(tmp_path / "no" / "such" / "dir.csv"), ["f.txt"], rows, [0, 1])
    with pytest.raises(OSError):
        w.finish()
    assert csvw.enabled()
    monkeypatch.setenv("FANDOM_SEARCH_NATIVE_CSV", "0")
    assert not csvw.enabled()
