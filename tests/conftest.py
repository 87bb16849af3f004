import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def synth_base():
    """Vocabulary words, embedding table and string table of SURVEY 8(d)."""
    from fandom_search_amd import synth, vocab
    words = synth.vocab_words()
    emb = synth.embedding()
    chars, off = vocab.pack_strings(words)
    return dict(words=words, emb=emb, chars=chars, off=off)
