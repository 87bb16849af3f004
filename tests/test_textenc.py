"""The native text front end (fs_textenc_*, fandom_search_amd/textenc.py) against the Python
path it stands in for (search.tokenize_files: read, chunk_text, tokenizer.tokenize): the same
tokens for every work -- plain words, punctuation and contractions (chunks the encoder has to
be taught), every kind of whitespace Python's str.split() knows, non-ASCII text, words the
vocabulary lacks, works of 100000 bytes and more, an empty work; and the same exceptions for
a missing file and for bytes that are not UTF-8.  No GPU involved."""

import os

import numpy as np
import pytest

from fandom_search_amd import search, synth, textenc, tokenizer, vocab


def _python_tokens(files, voc):
    out = []
    for f in files:
        out.append([t for t in search.read_work_tokens(f)])
    return out


def _native_tokens(files, voc, enc=None):
    enc = enc or textenc.TextEncoder(voc, threads=4)
    lens, sids = enc.encode_files(files)
    assert int(lens.sum()) == len(sids) and len(lens) == len(files)
    off = np.concatenate([[0], np.cumsum(lens)])
    return [[voc.strings[int(s)] for s in sids[int(off[i]):int(off[i + 1])]] for i in range(len(files))], enc


TEXTS = {
    "plain.txt": "baba babe babi\nbabo  babu\tbaca",
    "prose.txt": "\"Help me, Obi-Wan Kenobi. You're my only hope,\" she said -- twice!  Don't you think it's 3.5km (or $5)?",
    "spaces.txt": "a b c　d\x1ce\x85f g\r\nh\x0bi\x0cj  k l m",
    "unicode.txt": "café naïve 日本語 ¿qué? \U0001f600 smile﻿",
    "oov.txt": "Zorgblatt and Quuxly talk: zorgblatt, QUUXLY; baba-babe babe/babi e.g. U.S.A. www.example.com/x?y=1",
    "empty.txt": "",
    "onlyspace.txt": " \n\t  ",
}


@pytest.fixture()
def corpus(tmp_path):
    files = []
    for name, text in TEXTS.items():
        p = tmp_path / name
        p.write_bytes(text.encode("utf-8"))
        files.append(str(p))
    words = synth.vocab_words()
    rng = np.random.default_rng(3)
    for i in range(40):                               # enough files for several threads
        p = tmp_path / ("w%03d.txt" % i)
        ids = rng.integers(0, len(words), size=300 + i)
        toks = [words[int(t)] + ("," if j % 17 == 3 else "") for j, t in enumerate(ids)]
        p.write_text(" ".join(toks))
        files.append(str(p))
    long = tmp_path / "long.txt"                      # >= 100000 bytes: the chunked Python path
    long.write_text(" ".join(words[int(t)] for t in rng.integers(0, len(words), size=30000)))
    files.insert(3, str(long))
    return files


def test_native_tokens_equal_the_rule_tokenizers(corpus):
    assert textenc.enabled()
    words = synth.vocab_words()
    voc = vocab.Vocab(words, synth.embedding())
    want = _python_tokens(corpus, voc)
    got, enc = _native_tokens(corpus, voc)
    for f, a, b in zip(corpus, got, want):
        assert a == b, os.path.basename(f)
    assert sum(len(t) for t in want) > 40000 and any("," in t for t in want[-1])
    # a second pass: everything is known natively now (no placeholder), same tokens
    again, _ = _native_tokens(corpus, voc, enc)
    assert again == want
    # the background form
    enc.start(corpus[:10])
    lens, sids = enc.encode_files(corpus[:10])
    assert [len(t) for t in want[:10]] == lens.tolist()
    # ... and tokenize_files (the pool workers' form) agrees on the string ids' texts
    lens2, sids2, new = search.tokenize_files(corpus[:10], voc)
    assert lens2.tolist() == lens.tolist() and not new
    assert [voc.strings[int(s)] for s in sids2] == [voc.strings[int(s)] for s in sids]


def test_native_errors_are_the_python_paths(tmp_path):
    words = synth.vocab_words()
    voc = vocab.Vocab(words, synth.embedding())
    enc = textenc.TextEncoder(voc, threads=2)
    good = tmp_path / "good.txt"
    good.write_text("baba babe")
    with pytest.raises(FileNotFoundError):
        enc.encode_files([str(good), str(tmp_path / "missing.txt")])
    with pytest.raises(FileNotFoundError):
        search.read_work_tokens(str(tmp_path / "missing.txt"))
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"baba \xff\xfe babe")
    with pytest.raises(UnicodeDecodeError):
        enc.encode_files([str(good), str(bad)])
    with pytest.raises(UnicodeDecodeError):
        search.read_work_tokens(str(bad))
    trunc = tmp_path / "trunc.txt"
    trunc.write_bytes("café".encode("utf-8")[:-1])          # a sequence cut short
    with pytest.raises(UnicodeDecodeError):
        enc.encode_files([str(trunc)])


def test_switch(monkeypatch):
    monkeypatch.setenv("FANDOM_SEARCH_NATIVE_TEXT", "0")
    assert not textenc.enabled()
    monkeypatch.delenv("FANDOM_SEARCH_NATIVE_TEXT")
    monkeypatch.setenv("FANDOM_SEARCH_TOKENIZER", "simple")
    assert not textenc.enabled()


def test_a_known_batch_brings_its_vector_ids(corpus):
    """The encoding threads make what the search asks of a batch's tokens besides the string
    ids (fs_textenc_encode_files_vec) -- vector id per token, the number of out-of-vocabulary
    tokens, whether the two id arrays are equal -- when the encoder knows every chunk and no
    work is left to the Python path; otherwise (first sight of a chunk, a long work) the
    caller makes them as before."""
    voc = vocab.Vocab(synth.vocab_words(), synth.embedding())
    enc = textenc.TextEncoder(voc, threads=4)
    short = [f for f in corpus if not f.endswith("long.txt")]
    enc.start(short)
    lens0, tok0 = enc.encode_files(short)              # chunks it has to be taught: nothing made
    assert enc.last_vec is None
    enc.start(short)
    lens, tok = enc.encode_files(short)                # every chunk known now
    assert enc.last_vec is not None and enc.last_vec[0] is tok
    assert np.array_equal(lens, lens0) and np.array_equal(tok, tok0)
    _, vec, n_oov, same = enc.last_vec
    want = voc.vec_ids()[tok]
    assert vec.dtype == np.uint32 and np.array_equal(vec, want)
    assert n_oov == int(np.count_nonzero(want & np.uint32(vocab.OOV_FLAG))) and n_oov > 0
    assert same == bool(np.array_equal(tok, want)) and not same
    lens2, tok2 = enc.encode_files(short)              # not started: the same, on the caller's thread
    assert enc.last_vec is not None and np.array_equal(tok2, tok) and np.array_equal(enc.last_vec[1], want)
    enc.start(corpus)                                  # a work of 100000 bytes: the Python path, nothing made
    enc.encode_files(corpus)
    assert enc.last_vec is None
    # a batch whose string ids ARE its vector ids (plain words of the table only)
    plain = [f for f in short if os.path.basename(f) == "plain.txt"] * 8
    enc.encode_files(plain)
    enc.start(plain)
    _, tokp = enc.encode_files(plain)
    assert enc.last_vec is not None and enc.last_vec[3] == bool(np.array_equal(tokp, voc.vec_ids()[tokp]))
    assert enc.last_vec[2] == 0
