"""Fixtures the reference's OWN code produced (tests/golden/make_search_golden.py compiles
single definitions out of /root/reference/search.py with `ast` and runs them in the build
container): mk_vectors, sp_parse_chunks, write_records and new_record_structure on
duck-typed inputs, and the body of AnnIndexSearch.search around a given engine.  NearPy and
python-Levenshtein stay restated (parity of those unpinned); everything search.py itself
does is held to the reference's output here -- by the oracle, by the host code of the
product and (-m gpu) by the HIP path."""

import base64
import hashlib
import json
import os

import numpy as np
import pytest

from fandom_search_amd import search, synth, vocab
from tests import util

with open(os.path.join(util.GOLDEN, "search_golden.json"), encoding="utf-8") as _fh:
    G = json.load(_fh)


# ---- mk_vectors (search.py:65-84) -----------------------------------------------------------

def _hash_of(case):
    values = case["hash_values"]
    return (lambda s: values[s]) if case["hash"] == "crafted" else vocab.default_oov_hash


@pytest.mark.parametrize("name", sorted(G["mk_vectors"]["cases"]))
def test_oracle_mk_vectors_is_the_references(name):
    from oracle import search_restated as sr
    g = G["mk_vectors"]
    case = g["cases"][name]
    table = {w: np.asarray(v, dtype=np.float32) for w, v in g["table"].items()}
    toks = [sr.Tok(w, 0, w, 0, table.get(w, np.zeros(g["dim"], np.float32)), w in table)
            for w in case["words"]]
    got = sr.mk_vectors(toks, _hash_of(case))
    assert list(got.shape) == case["shape"] and got.dtype == np.float64
    assert got.tolist() == case["vectors"]
    # the seeded hash the fixture was made with is the product's default
    if case["hash"] == "seeded":
        for s, h in case["hash_values"].items():
            assert vocab.default_oov_hash(s) == h


@pytest.mark.parametrize("name", sorted(G["mk_vectors"]["cases"]))
def test_host_vector_ids_carry_the_references_vectors(name):
    """The product never builds mk_vectors' matrix: a token is a vector id (table row, or
    OOV_FLAG | the three hot positions).  Decoded again, every id is the row the reference
    made -- also where two or three of the hot positions coincide."""
    g = G["mk_vectors"]
    case = g["cases"][name]
    words = list(g["table"])
    voc = vocab.Vocab(words, np.asarray([g["table"][w] for w in words], dtype=np.float32),
                      oov_hash=_hash_of(case))
    sids, vids = voc.encode(case["words"])
    assert len(vids) == len(case["words"])
    for sid, vid, want in zip(sids, vids, case["vectors"]):
        assert voc.vector(int(sid)).astype(np.float64).tolist() == want
        assert bool(int(vid) & vocab.OOV_FLAG) == (voc.strings[int(sid)] not in g["table"])
    # equal vectors <=> equal ids (what the exact pipeline relies on)
    for i in range(len(vids)):
        for j in range(len(vids)):
            assert (vids[i] == vids[j]) == (case["vectors"][i] == case["vectors"][j])


# ---- sp_parse_chunks (search.py:47-63) ------------------------------------------------------

@pytest.mark.parametrize("name", sorted(G["sp_parse_chunks"]))
def test_chunk_text_cuts_where_the_reference_cuts(name):
    from tests.golden.make_search_golden import chunk_text_recipe
    case = G["sp_parse_chunks"][name]
    text = chunk_text_recipe(case["kind"], case["length"], case["seed"])
    assert hashlib.sha256(text.encode("utf-8")).hexdigest() == case["sha256"]
    if case["error"] == "IndexError":
        # (the reference's own: a text of exactly 100000 characters, or no space at all)
        with pytest.raises(IndexError):
            list(vocab.chunk_text(text))
    elif case["error"] == "endless":
        # the reference yields chunks for ever on this text; the product says so instead
        with pytest.raises(ValueError, match="100000"):
            list(vocab.chunk_text(text))
    else:
        assert [text[a:b] for a, b in case["chunks"]] == list(vocab.chunk_text(text))


# ---- write_records / new_record_structure (search.py:20-37, 331-334) ------------------------

def _records():
    def dec(x):
        return float(x["float"]) if isinstance(x, dict) else x
    return [[dec(x) for x in r] for r in G["write_records"]["records"]]


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_write_records_bytes_are_the_references(tmp_path, which):
    if which == "product":
        write = search.write_records
    else:
        from oracle import search_restated as sr
        write = sr.write_records
    g = G["write_records"]
    p = str(tmp_path / "out.csv")
    write(_records(), p)
    assert open(p, "rb").read() == base64.b64decode(g["csv_base64"])
    write([search.new_record_structure['fields']] + _records()[:1], p)
    assert open(p, "rb").read() == base64.b64decode(g["header_and_first_base64"])
    write([], p)
    assert open(p, "rb").read() == base64.b64decode(g["empty_base64"])


def test_record_structure_is_the_references():
    g = G["write_records"]
    assert search.new_record_structure['fields'] == g["fields"]
    assert [t.__name__ for t in search.new_record_structure['types']] == g["types"]
    from oracle import search_restated as sr
    assert sr.FIELDS == g["fields"]


# ---- the body of AnnIndexSearch.search (search.py:163-226) ----------------------------------

def test_golden_csvs_are_what_the_references_search_body_produced():
    """make_search_golden.py ran the reference's search body on the inputs of every golden
    case around the restated engine and compared with the committed CSVs before recording
    their digests: the committed files are still those."""
    seen = set()
    for entry in G["search_body_on_golden_cases"]:
        text = util.golden_text(entry["case"], entry["tag"])
        assert hashlib.sha256(text.encode("utf-8")).hexdigest() == entry["sha256"]
        assert text.count("\r\n") == entry["records"]
        seen.add((entry["case"], entry["tag"]))
    assert seen == {(c, t) for c in util.GOLDEN_CASES for t in ("canonical", "literal")}


def _mixed_inputs():
    g = G["search_body_mixed"]
    words = synth.vocab_words()
    emb = synth.embedding()
    script = np.asarray(g["script"], dtype=np.uint32)
    assert script.tolist() == synth.script_tokens(len(script)).tolist()
    return g, words, emb, script


def test_oracle_search_body_is_the_references_on_the_mixed_case():
    """is_space tokens, capitalised and out-of-vocabulary fan words, a work shorter than a
    window and an empty one: the restatement's records = the reference's."""
    from oracle import nearpy_restated as nr
    from oracle import search_restated as sr
    from tests.golden.make_search_golden import to_csv
    g, words, emb, script = _mixed_inputs()
    voc = vocab.Vocab(words, emb)
    scene, char = synth.script_columns(len(script))
    rows = [[words[t], voc.orth(t), int(scene[i]), char[i]] for i, t in enumerate(script)]
    stoks = [sr.Tok(words[i], voc.orth(i), words[i], voc.orth(i), emb[i]) for i in script]
    idx = sr.AnnIndexSearch(rows, stoks, g["window_size"], 15, 14, 0.1, synth.lsh_normals(g["window_size"]),
                            oov_hash=vocab.default_oov_hash, arith=nr.CanonicalArith(), unique_filter=True)
    out = []
    for name, text in g["files"]:
        fan = []
        for w in text.split():                         # (whitespace tokens dropped, search.py:166)
            sid = voc.string_id(w)
            vec = voc.vector(sid)
            fan.append(sr.Tok(w, voc.orth(sid), w.lower(), 0, vec, voc.has_vector(sid)))
        out += idx.search(name, fan)
    assert to_csv(out) == g["csv"]
    assert idx.windows_processed == g["windows_processed"]


@pytest.mark.gpu
def test_hip_search_is_the_references_on_the_mixed_case(tmp_path, monkeypatch):
    """The product end to end at the class seam (script file, fan files, tokenizer, vector
    ids with OOV 3-hot codes, HIP search, record join, csv writer) against the CSV the
    reference's own search body wrote for the same files."""
    g, words, emb, script = _mixed_inputs()
    monkeypatch.chdir(tmp_path)
    (tmp_path / "script.txt").write_text(synth.script_markup(script, words))
    for name, text in g["files"]:
        (tmp_path / name).write_text(text, encoding="utf8")
    voc = vocab.Vocab(words, emb)
    ann = search.AnnIndexSearch(str(tmp_path / "script.txt"), g["window_size"], 15, 14, 0.1,
                                vocab=voc, normals=synth.lsh_normals(g["window_size"]), unique_filter=True)
    recs = []
    for name, _ in g["files"]:
        recs += ann.search(name)
    search.write_records(recs, "got.csv")
    assert (tmp_path / "got.csv").read_bytes() == g["csv"].encode("utf-8")
    assert ann.windows_processed == g["windows_processed"]


# ---- the driver: analyze (search.py:336-399) ------------------------------------------------

class _CannedSearcher(object):
    """Stands where AnnIndexSearch stands in search.analyze: the records the fixture's stand-in
    returned per file name, as the numeric rows + fan words the product's searcher hands over."""

    def __init__(self, script_words):
        from fandom_search_amd import abi
        from tests.golden.make_search_golden import canned_records
        self._canned, self._abi, self._words = canned_records, abi, script_words
        self.word_lowercase = tuple(script_words)
        self.orth_id = tuple(vocab.hash_string(w) for w in script_words)
        self.character = tuple("CHAR%d" % (o % 3) for o in range(len(script_words)))
        self.scene = tuple(o // 4 for o in range(len(script_words)))

    def search_rows(self, filenames):
        rows, words = [], []
        for w, f in enumerate(filenames):
            for r in self._canned(f, self._words):
                rows.append((w, r[1], r[4], r[10], r[9], r[11]))
                words.append(r[2])
        return np.array(rows, dtype=self._abi.ROW_DTYPE), words


@pytest.mark.parametrize("label", [c["label"] for c in G["analyze"]["cases"]])
def test_analyze_writes_what_the_references_driver_writes(tmp_path, monkeypatch, capsys, label):
    """Listing, seeded shuffle, -s / -n window, clusters, batch files, the dated file and its
    -k suffix, the printed lines: search.analyze against the reference's own analyze, which
    make_search_golden.py ran around a stand-in index and pool (the listing order -- sorted --
    and today's date are inputs there and here)."""
    import base64 as b64
    import datetime as real_datetime
    import types
    g = G["analyze"]
    case = next(c for c in g["cases"] if c["label"] == label)
    monkeypatch.chdir(tmp_path)
    (tmp_path / "fan").mkdir()
    for name in g["names"]:
        (tmp_path / "fan" / name).write_text("x")
    for name in case["pre_existing"]:
        (tmp_path / name).write_text("old\n")
    fixed = types.SimpleNamespace(date=types.SimpleNamespace(today=lambda: real_datetime.date(2020, 2, 29)))
    monkeypatch.setattr(search, "datetime", fixed)
    args = types.SimpleNamespace(fan_works="fan", script="script.txt", skip_works=case["skip_works"],
                                 num_works=case["num_works"])
    search.analyze(args, chunk_size=case["chunk_size"], searcher=_CannedSearcher(g["script_words"]))
    assert capsys.readouterr().out == case["stdout"]
    got = {f: (tmp_path / f).read_bytes() for f in sorted(os.listdir(tmp_path)) if (tmp_path / f).is_file()}
    want = {f: b64.b64decode(v) for f, v in case["files"].items()}
    assert sorted(got) == sorted(want)
    for f in want:
        assert got[f] == want[f], f
    # (the reference's pool call, for the record: Pool(4, maxtasksperchild=10).map(..., chunksize = chunk_size // 16))
    pools = [c for c in case["calls"] if "processes" in c]
    assert all(c["processes"] == 4 and c["maxtasksperchild"] == 10 and c["chunksize"] == case["chunk_size"] // 16
               for c in pools)


# ---- the script parser: load_markup_script (search.py:290-329) ------------------------------

@pytest.mark.parametrize("name", sorted(G["load_markup_script"]))
def test_load_markup_script_rows_are_the_references(tmp_path, capsys, name):
    """Regex precedence per text line, scene numbers (digits only), the permanent fall-back to
    the running count after the first label without digits, the lines it prints: the rows the
    reference's own parser made (around a whitespace splitter; the words of these scripts are
    plain, so the rule tokenizer splits them the same way)."""
    case = G["load_markup_script"][name]
    p = tmp_path / "script.txt"
    p.write_bytes(case["markup"].encode("utf-8"))
    rows = search.load_markup_script(str(p))
    assert capsys.readouterr().out == case["stdout"]
    assert rows == case["rows"]
