"""Hypothesis-driven parity: random small problems (tests/fuzzcase.py).
CPU: the plain-C oracle against the literal Python oracle (canonical mode).
GPU: the HIP library against the plain-C oracle, whichever pipeline the index
selects."""

import numpy as np
import pytest
import os

from hypothesis import HealthCheck, given, settings, strategies as st

from fandom_search_amd import abi
from fandom_search_amd.vocab import oov_vector
from tests import fuzzcase, util

CASE = st.fixed_dictionaries(dict(
    seed=st.integers(0, 2 ** 31 - 1),
    n=st.integers(1, 12),
    H=st.integers(1, 6),
    B=st.integers(1, 6),
    D=st.sampled_from([4, 9, 16]),
    V=st.integers(3, 24),
    unique=st.booleans(),
    thr=st.sampled_from([0.02, 0.1, 0.3]),
    one_hot=st.booleans(),
    oov_rate=st.sampled_from([0.0, 0.0, 0.08]),
    n_script=st.integers(0, 70),
    works=st.lists(st.integers(0, 60), min_size=0, max_size=4),
))


def _c_oracle(case):
    from oracle import c_oracle
    from fandom_search_amd.vocab import pack_strings
    sch, so = pack_strings(case["swords"])
    oi = c_oracle.OracleIndex(case["cfg"], case["script"], sch, so, case["emb"], case["normals"],
                              threads=2)
    return oi.search(case["tok"], case["off"], case["chars"], case["coff"],
                     tok_str=case["tok_str"])


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(CASE)
def test_c_oracle_equals_python_oracle(p):
    from oracle import nearpy_restated as nr
    from oracle import search_restated as sr
    case = fuzzcase.make_case(**p)
    cfg, emb, strings = case["cfg"], case["emb"], case["strings"]
    D = cfg.emb_dim

    def toks(vecs, sids, lower=False):
        out = []
        for v, s in zip(vecs, sids):
            text = strings[int(s)].lower() if lower else strings[int(s)]
            vec = oov_vector(int(v), D) if int(v) & abi.FS_OOV_FLAG else emb[int(v)]
            out.append(sr.Tok(text, int(s), text.lower(), int(s), vec))
        return out

    script_sid = [strings.index(w) if w in strings else 0 for w in case["swords"]]
    script_toks = [sr.Tok(w, i, w, i, oov_vector(int(v), D) if int(v) & abi.FS_OOV_FLAG
                          else emb[int(v)])
                   for i, (w, v) in enumerate(zip(case["swords"], case["script"]))]
    rows = [[w, 0, 0, "C"] for w in case["swords"]]
    want, st_ = _c_oracle(case)
    if len(script_toks) == 0:
        assert len(want) == 0
        return
    idx = sr.AnnIndexSearch(rows, script_toks, cfg.window_size, cfg.number_of_hashes,
                            cfg.hash_dimensions, cfg.distance_threshold,
                            [case["normals"][h] for h in range(cfg.number_of_hashes)],
                            arith=nr.CanonicalArith(), unique_filter=bool(cfg.unique_filter))
    py = []
    off = case["off"]
    for w in range(len(off) - 1):
        lo, hi = int(off[w]), int(off[w + 1])
        py += idx.search(w, toks(case["tok"][lo:hi], case["tok_str"][lo:hi]))
    assert len(py) == len(want)
    for a, b in zip(py, want):
        assert (a[0], a[1], a[4], a[10]) == (b["work"], b["fan_ix"], b["orig_ix"], b["lev"])
        assert a[9] == b["dist"] and a[11] == b["comb"]


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("FS_FUZZ_EXAMPLES", "400")), deadline=None,
          suppress_health_check=list(HealthCheck))
@given(CASE)
def test_hip_equals_c_oracle(p):
    from fandom_search_amd.engine import ScriptIndex
    case = fuzzcase.make_case(**p)
    want, ost = _c_oracle(case)
    ix = ScriptIndex(case["script"], case["swords"], case["emb"], case["normals"], cfg=case["cfg"])
    got, st_ = ix.search(ix.corpus(case["tok"], case["off"], case["chars"], case["coff"],
                                   tok_str=case["tok_str"]))
    util.assert_rows_equal(got, want)
    assert st_.matches == ost.matches and st_.windows_processed == ost.windows_processed
    ix.close()
