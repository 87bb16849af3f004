"""GPU parity on a vector table of the size and shape of a real word-embedding
model (spaCy en_core_web_md keeps 20k distinct rows, _lg 685k; rows are not
unit length and near-synonyms sit close together): 200 000 rows with norms in
[2, 8] and clusters of four near-synonyms.  The exact n-gram proof cannot hold
(c_max ~ 1), so the index takes the LSH pipeline with its per-index tables
(projection tables 3 GB, pair table ~2.5 GB) -- a check of 64-bit indexing and of
the distance bounds on unnormalised vectors against the plain-C oracle."""

import numpy as np
import pytest

from fandom_search_amd import abi, synth
from fandom_search_amd.vocab import pack_strings
from tests import util

pytestmark = pytest.mark.gpu

V, D = 200_000, 300


def _table():
    rng = np.random.default_rng(77)
    centers = rng.standard_normal((V // 4, D)).astype(np.float32)
    emb = np.repeat(centers, 4, axis=0)
    emb += 0.2 * rng.standard_normal((V, D)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    emb *= rng.uniform(2.0, 8.0, size=(V, 1)).astype(np.float32)
    perm = rng.permutation(V)
    return np.ascontiguousarray(emb[perm]), perm, np.argsort(perm)


def test_large_unnormalised_table_matches_oracle():
    from fandom_search_amd.engine import ScriptIndex
    from oracle import c_oracle
    emb, perm, inv = _table()
    rng = np.random.default_rng(78)
    n_script = 6000
    # script over the whole id range, high ids included
    script = np.concatenate([rng.integers(0, V, size=n_script - 2),
                             [V - 1, 0]]).astype(np.uint32)
    works = []
    for w in range(24):
        t = rng.integers(0, V, size=900).astype(np.uint32)
        for _ in range(4):                        # planted script spans
            ln = int(rng.integers(6, 25))
            a = int(rng.integers(0, n_script - ln))
            b = int(rng.integers(0, len(t) - ln))
            t[b:b + ln] = script[a:a + ln]
        sel = np.nonzero(rng.random(len(t)) < 0.08)[0]     # synonyms swapped in
        syn = (inv[t[sel]] // 4) * 4 + rng.integers(0, 4, size=len(sel))
        t[sel] = perm[syn].astype(np.uint32)
        works.append(t)
    off = np.zeros(len(works) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(t) for t in works])
    tok = np.concatenate(works)
    # strings: one per distinct id in use (string ids are separate from vector ids)
    used, tok_str = np.unique(np.concatenate([tok, script]), return_inverse=True)
    names = ["t%x" % int(u) for u in used]
    chars, coff = pack_strings(names)
    tok_str = tok_str.astype(np.uint32)
    swords = [names[i] for i in tok_str[len(tok):]]
    tok_str = np.ascontiguousarray(tok_str[:len(tok)])
    normals = synth.lsh_normals(6)
    cfg = abi.make_config()
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    assert ix.info["proof_ok"] == 0
    got, st = ix.search(ix.corpus(tok, off, chars, coff, tok_str=tok_str))
    assert st.path == abi.FS_MODE_GENERAL
    sch, so = pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=8)
    want, ost = oi.search(tok, off, chars, coff, tok_str=tok_str)
    util.assert_rows_equal(got, want)
    assert st.matches == ost.matches
    # verbatim spans and synonym-swapped spans both produce rows
    assert len(got) > 200 and int((got["dist"] > 1e-9).sum()) > 0
    ix.close()
