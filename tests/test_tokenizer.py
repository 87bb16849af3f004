"""The restated spaCy tokenizer (fandom_search_amd/tokenizer.py) against the examples of
spaCy's own documentation and the behaviours its English rules are known for.  The
reference tokenises with spaCy (/root/reference/search.py:47-63, 166, 322-323); spaCy is
not installed here, so these expectations are hand-derived from the published rules
(usage/linguistic-features "How spaCy's tokenizer works", usage/spacy-101)."""

import pytest

from fandom_search_amd import tokenizer, vocab


@pytest.mark.parametrize("text,want", [
    # the worked example of the tokenizer documentation
    ("Let's go to N.Y.!", ["Let", "'s", "go", "to", "N.Y.", "!"]),
    ('"Let\'s go!"', ['"', "Let", "'s", "go", "!", '"']),
    # spaCy 101
    ("Apple is looking at buying U.K. startup for $1 billion",
     ["Apple", "is", "looking", "at", "buying", "U.K.", "startup", "for", "$", "1", "billion"]),
    # contractions are tokenizer exceptions
    ("don't", ["do", "n't"]), ("Don't", ["Do", "n't"]), ("can't", ["ca", "n't"]),
    ("won't", ["wo", "n't"]), ("isn't", ["is", "n't"]), ("cannot", ["can", "not"]),
    ("I'm", ["I", "'m"]), ("we're", ["we", "'re"]), ("it's", ["it", "'s"]), ("its", ["its"]),
    ("I'll've", ["I", "'ll", "'ve"]), ("gonna", ["gon", "na"]), ("he'd", ["he", "'d"]),
    ("well", ["well"]), ("hell", ["hell"]), ("were", ["were"]),
    # possessive suffix, hyphen infix between letters
    ("Obi-Wan Kenobi's lightsaber", ["Obi", "-", "Wan", "Kenobi", "'s", "lightsaber"]),
    ("well-known", ["well", "-", "known"]),
    # abbreviations keep their period; a sentence-final period is split off
    ("Mr. Smith", ["Mr.", "Smith"]), ("e.g.", ["e.g."]), ("a.m.", ["a.m."]), ("U.S.", ["U.S."]),
    ("UK.", ["UK", "."]), ("end.", ["end", "."]), ("No.", ["No", "."]),
    # numbers: decimals and thousands stay, units and currency split
    ("3.5", ["3.5"]), ("1,000", ["1,000"]), ("10km", ["10", "km"]), ("$5", ["$", "5"]),
    ("50%", ["50", "%"]), ("9am", ["9", "am"]),
    # brackets, quotes, ellipses
    ("(hello)", ["(", "hello", ")"]), ("hello...", ["hello", "..."]),
    ('She said, "no."', ["She", "said", ",", '"', "no", ".", '"']),
    # URLs and emoticons are single tokens
    ("http://example.com/a?b=c is a URL.", ["http://example.com/a?b=c", "is", "a", "URL", "."]),
    (":-) C++", [":-)", "C++"]),
    # comma between letters, '=' between letters
    ("a,b", ["a", ",", "b"]), ("x=y", ["x", "=", "y"]),
])
def test_known_answers(text, want):
    assert tokenizer.tokenize(text) == want


def test_whitespace_runs_produce_no_tokens():
    """Runs of whitespace would be is_space tokens, which the reference drops."""
    assert tokenizer.tokenize("  one\t two \n\n three  ") == ["one", "two", "three"]
    assert tokenizer.tokenize("") == []


def test_vocab_tokenize_defaults_to_the_rules(monkeypatch):
    monkeypatch.delenv("FANDOM_SEARCH_TOKENIZER", raising=False)
    assert vocab.tokenize("Don't panic!") == ["Do", "n't", "panic", "!"]
    monkeypatch.setenv("FANDOM_SEARCH_TOKENIZER", "simple")
    assert vocab.tokenize("Don't panic!") == ["Don", "'", "t", "panic", "!"]


def test_synthetic_text_is_split_on_spaces_only():
    """Corpora of space-separated alphabetic words: every tokenizer gives the words."""
    from fandom_search_amd import synth
    words = synth.vocab_words()[:500]
    text = " ".join(words)
    assert tokenizer.tokenize(text) == words
