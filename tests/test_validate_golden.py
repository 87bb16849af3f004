"""`validate` (search.py:228-285) against fixtures the reference itself produced
(tests/golden/make_validate_golden.py executes the reference's function in the build
container; only its inputs and outputs are committed): printed report, prompts, answers
consumed and return value, case by case."""

import builtins
import contextlib
import io
import json
import os

import pytest

from fandom_search_amd import search

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "validate_golden.json"), encoding="utf-8") as _fh:
    GOLDEN = json.load(_fh)


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_validate_matches_the_reference(name, tmp_path, monkeypatch):
    case = GOLDEN[name]
    path = tmp_path / "script.txt"
    with open(path, "w", encoding="utf-8", newline="") as fh:
        fh.write(case["script"])
    out = io.StringIO()
    left = list(case["answers"])

    def fake_input(prompt=""):
        out.write(prompt)
        out.write("\n")
        return left.pop(0)

    monkeypatch.setattr(builtins, "input", fake_input)
    with contextlib.redirect_stdout(out):
        if case["interactive"]:
            result = search.validate_markup_script(str(path), True)
        else:
            result = search.validate_markup_script(str(path))
    assert out.getvalue() == case["stdout"]
    assert result is case["returns"]
    assert len(case["answers"]) - len(left) == case["answers_used"]


def test_fixture_covers_every_branch():
    """Clean and failing scripts, each of the three reports, the interactive prompt with
    every kind of answer."""
    text = "".join(c["stdout"] for c in GOLDEN.values())
    for needle in ("No markup errors found.", "Unbalanced left tag delimiters:",
                   "Unbalanced right tag delimiters:", "Unexpected tag labels:",
                   "Do you want to continue?", "Enter y for yes or n for no: "):
        assert needle in text
    assert {c["returns"] for c in GOLDEN.values()} == {True, False}
    assert len(GOLDEN) >= 20
