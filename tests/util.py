"""Shared helpers of the test-suite (inputs and comparisons)."""

import numpy as np

from fandom_search_amd import abi, synth


def ragged_corpus(lengths, script, first_work=0):
    """Packed corpus whose work w has lengths[w] tokens (0 allowed)."""
    parts = [synth.fanwork_tokens(first_work + i, int(n), script) if n else
             np.zeros(0, dtype=np.uint32) for i, n in enumerate(lengths)]
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(p) for p in parts])
    tok = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)
    return tok.astype(np.uint32), off


def assert_rows_equal(got, want):
    """Bit-exact comparison of two abi.ROW_DTYPE arrays."""
    assert len(got) == len(want), (len(got), len(want))
    for name in abi.ROW_DTYPE.names:
        a, b = got[name], want[name]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint64), b.view(np.uint64)
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, (name, int(bad[0]), got[bad[0]], want[bad[0]])


def oracle_index(cfg, script, words, emb, normals, threads=8, swords=None):
    from oracle import c_oracle
    from fandom_search_amd.vocab import pack_strings
    sch, so = pack_strings(swords if swords is not None else [words[int(t)] for t in script])
    return c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads)


# ---- golden fixtures --------------------------------------------------------

import csv
import io
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ("synthetic_small", "synthetic_n4", "crowded_unique", "crowded_nounique")


def load_case(name):
    with open(os.path.join(GOLDEN, name + ".json")) as fh:
        return json.load(fh)


def golden_text(name, tag):
    with open(os.path.join(GOLDEN, "%s.%s.csv" % (name, tag)), newline="") as fh:
        return fh.read()


def case_arrays(case):
    works = case["works"]
    off = np.zeros(len(works) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(w) for w in works])
    tok = np.asarray([t for w in works for t in w], dtype=np.uint32)
    return tok, off


def case_config(case, **kw):
    return abi.make_config(window_size=case["window_size"],
                           number_of_hashes=case["number_of_hashes"],
                           hash_dimensions=case["hash_dimensions"],
                           distance_threshold=case["distance_threshold"],
                           unique_filter=case["unique_filter"], **kw)


def rows_to_csv(rows, case, words):
    """Join fs_row records with words / orth ids / character / scene and
    serialise like write_records (search.py:331-334)."""
    from fandom_search_amd.vocab import hash_string
    script = case["script"]
    scene, char = synth.script_columns(len(script))
    out = []
    for r in rows:
        w, f, o = int(r["work"]), int(r["fan_ix"]), int(r["orig_ix"])
        fw = words[case["works"][w][f]]
        ow = words[script[o]]
        out.append([synth.work_name(w), f, fw, hash_string(fw), o, ow, hash_string(ow),
                    char[o], int(scene[o]), float(r["dist"]), int(r["lev"]), float(r["comb"])])
    buf = io.StringIO()
    csv.writer(buf).writerows(out)
    return buf.getvalue()
