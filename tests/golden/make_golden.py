#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ with the literal Python
oracle (oracle/search_restated.py + oracle/nearpy_restated.py).

The reference itself cannot run in this image (nearpy / spacy / Levenshtein
are absent), so these vectors come from the restatement, not
from the reference: PARITY UNPINNED.  Each case is stored as
  <case>.json           the inputs (token ids, parameters, seeds)
  <case>.canonical.csv  records in the canonical arithmetic (bit-exact target
                        of the C oracle and the HIP library)
  <case>.literal.csv    records with numpy/BLAS arithmetic as NearPy would run
                        (distances compared with 1e-12 tolerance)

Run from the repo root:  python tests/golden/make_golden.py
"""

import csv
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from fandom_search_amd import synth, vocab  # noqa: E402
from oracle import nearpy_restated as nr  # noqa: E402
from oracle import search_restated as sr  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run_case(case, arith):
    words = synth.vocab_words()
    emb = synth.embedding()
    voc = vocab.Vocab(words, emb)
    n = case["window_size"]
    normals = synth.lsh_normals(n, case["number_of_hashes"], case["hash_dimensions"])
    script = case["script"]
    scene, char = synth.script_columns(len(script))

    def toks(ids):
        return [sr.Tok(words[i], voc.orth(i), words[i], voc.orth(i), emb[i]) for i in ids]

    rows = [[words[t], voc.orth(t), int(scene[i]), char[i]] for i, t in enumerate(script)]
    idx = sr.AnnIndexSearch(rows, toks(script), n, case["number_of_hashes"],
                            case["hash_dimensions"], case["distance_threshold"], normals,
                            arith=arith, unique_filter=case["unique_filter"])
    out = []
    for w, ids in enumerate(case["works"]):
        out += idx.search(synth.work_name(w), toks(ids))
    return out


def to_csv(records):
    buf = io.StringIO()
    csv.writer(buf).writerows(records)
    return buf.getvalue()


def synthetic_case(n_works, tokens, script_tokens, n=6, unique=True):
    script = synth.script_tokens(script_tokens)
    works = [synth.fanwork_tokens(w, tokens, script).tolist() for w in range(n_works)]
    return dict(window_size=n, number_of_hashes=15, hash_dimensions=14,
                distance_threshold=0.1, unique_filter=unique,
                script=script.tolist(), works=works)


def crowded_case(unique):
    """A phrase that occurs 13 times in the script (NearestFilter(10) keeps the
    first ten), overlapping matches, a near miss and works shorter than n."""
    rng = np.random.default_rng(7)
    phrase = [11, 22, 33, 44, 55, 66, 77]
    script = []
    for _ in range(13):
        script += phrase + rng.integers(100, 8000, size=9).tolist()
    script += rng.integers(100, 8000, size=40).tolist()
    w0 = rng.integers(100, 8000, size=30).tolist() + phrase + rng.integers(100, 8000, size=20).tolist()
    w1 = script[3:40] + [5] + script[41:70]            # long verbatim span with one substitution
    w2 = phrase[:5]                                     # shorter than a window
    w3 = []
    w4 = phrase + phrase + script[100:112]
    return dict(window_size=6, number_of_hashes=15, hash_dimensions=14,
                distance_threshold=0.1, unique_filter=unique,
                script=[int(x) for x in script],
                works=[[int(x) for x in w] for w in (w0, w1, w2, w3, w4)])


CASES = {
    "synthetic_small": lambda: synthetic_case(8, 260, 700),
    "synthetic_n4": lambda: synthetic_case(5, 200, 500, n=4),
    "crowded_unique": lambda: crowded_case(True),
    "crowded_nounique": lambda: crowded_case(False),
}


def main():
    for name, make in CASES.items():
        case = make()
        with open(os.path.join(HERE, name + ".json"), "w") as fh:
            json.dump(case, fh)
        for tag, arith in (("canonical", nr.CanonicalArith()), ("literal", nr.LiteralArith())):
            recs = run_case(case, arith)
            with open(os.path.join(HERE, "%s.%s.csv" % (name, tag)), "w", newline="") as fh:
                fh.write(to_csv(recs))
            print(name, tag, len(recs), "records")


if __name__ == "__main__":
    main()
