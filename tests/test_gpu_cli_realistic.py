"""`ao3.py search` end to end in the regime of a real vector table: near-synonym
vectors (no exact-scan proof -> LSH pipeline), mixed-case text with punctuation
(the fan side is case-sensitive, the script side lower-cased: search.py:151 vs
:166), out-of-vocabulary words on both sides (3-hot vectors, search.py:79-83).
The CSV must equal the literal Python oracle's records for the same tokens."""

import datetime
import io
import csv

import numpy as np
import pytest

from fandom_search_amd import search, vocab
from fandom_search_amd.cli import main

pytestmark = pytest.mark.gpu

SCRIPT = """SCENE_NUMBER<<1>>
CHARACTER_NAME<<LEIA>>
LINE<<Help me, Obi-Wan Kenobi. You're my only hope.>>
DIRECTION<<She vanishes.>>
CHARACTER_NAME<<LUKE>>
LINE<<I have a very bad feeling about this, Artoo!>>
SCENE_NUMBER<<Two>>
CHARACTER_NAME<<HAN>>
LINE<<Never tell me the odds. Great shot kid, that was one in a million.>>
CHARACTER_NAME<<LEIA>>
LINE<<Someone has to save our skins. Into the garbage chute, flyboy!>>
"""

FANWORKS = {
    "a.txt": "Rey whispered: help me, Obi-Wan Kenobi. You're my only hope -- she said it twice. "
             "Help me Obi Wan Kenobi you are my only hope.",
    "b.txt": "I HAVE a very bad feeling about this, Artoo! said Poe. I have a really bad feeling "
             "about that, Artoo.",
    "c.txt": "Nothing from the film here, only Zorgblatt and Quuxly talking about tea.",
    "d.txt": "",
    "e.txt": "Never tell me the odds! Great shot kid, that was one in a million. Someone has to "
             "save our skins; into the garbage chute, flyboy. never tell me the chances, kid.",
}


def _table():
    """A 64-d table over the words of the script and a few more; synonyms get
    nearly parallel vectors ('bad'~'terrible', 'odds'~'chances', 'very'~'really')."""
    rng = np.random.default_rng(42)
    words = sorted(set(w.lower() for w in vocab.tokenize(SCRIPT.replace("<<", " ").replace(">>", " "))
                       if w.isalpha()) | {"really", "terrible", "chances", "that", "said", "she"})
    words = [w for w in words if w not in ("artoo", "kenobi", "flyboy")]      # stay OOV
    emb = rng.standard_normal((len(words), 64)).astype(np.float32)
    ix = {w: i for i, w in enumerate(words)}
    for a, b in (("bad", "terrible"), ("odds", "chances"), ("very", "really"), ("this", "that")):
        emb[ix[b]] = emb[ix[a]] + 0.1 * rng.standard_normal(64).astype(np.float32)
    # also give the capitalised / upper-case spellings that occur their own (same) vectors
    extra = {"Help": "help", "You": "you", "I": "i", "HAVE": "have", "Never": "never",
             "Great": "great", "Someone": "someone"}
    all_words = words + list(extra)
    emb = np.vstack([emb, np.stack([emb[ix[v]] for v in extra.values()])])
    return all_words, emb


def test_search_cli_with_synonyms_case_and_oov(tmp_path, monkeypatch, capsys):
    from oracle import nearpy_restated as nr
    from oracle import search_restated as sr
    words, emb = _table()
    np.savez(tmp_path / "vectors.npz", words=np.array(words), vectors=emb)
    monkeypatch.setenv("FANDOM_SEARCH_VECTORS", str(tmp_path / "vectors.npz"))
    search.set_vocab(None)
    (tmp_path / "script.txt").write_text(SCRIPT)
    fandir = tmp_path / "fan"
    fandir.mkdir()
    for name, text in FANWORKS.items():
        (fandir / name).write_text(text)
    monkeypatch.chdir(tmp_path)
    try:
        assert main(["search", str(fandir), str(tmp_path / "script.txt"), "--window-size", "4"]) == 0
        voc = search.get_vocab()
    finally:
        search.set_vocab(None)
    capsys.readouterr()
    today = '{:%Y%m%d}'.format(datetime.date.today())
    with open(tmp_path / ("match-4gram-%s.csv" % today), newline="") as fh:
        got = fh.read()

    # the literal oracle over the same tokens, vectors, hyperplanes and OOV hash
    n = 4
    rows = search.load_markup_script(str(tmp_path / "script.txt"))[1:]

    def tok(text, lower=False):
        t = text.lower() if lower else text
        sid = voc.string_id(t)
        return sr.Tok(t, voc.orth(sid), t.lower(), vocab.hash_string(t.lower()),
                      voc.vector(sid), has_vector=True)     # OOV vectors already materialised

    script_toks = [tok(r[0]) for r in rows]
    normals = search.default_normals(n, 15, 14, 64)
    idx = sr.AnnIndexSearch(rows, script_toks, n, 15, 14, 0.1, [normals[h] for h in range(15)],
                            arith=nr.CanonicalArith())
    want = [search.new_record_structure['fields']]
    for f in search.list_fan_works(str(fandir)):
        want += idx.search(f, [tok(t) for t in search.read_work_tokens(f)])
    buf = io.StringIO()
    csv.writer(buf).writerows(want)
    assert got == buf.getvalue()
    body = list(csv.reader(io.StringIO(got)))[1:]
    assert len(body) > 20
    # approximate matches through synonyms are present, exact ones have distance ~ 0
    dists = [float(r[9]) for r in body]
    assert any(d > 1e-6 for d in dists) and any(abs(d) < 1e-12 for d in dists)
    # a record on an out-of-vocabulary script word, and case kept on the fan side
    assert any(r[5] == "kenobi" for r in body)
    assert any(r[2] == "HAVE" for r in body)
