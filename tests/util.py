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


def oracle_index(cfg, script, words, emb, normals, threads=8):
    from oracle import c_oracle
    from fandom_search_amd.vocab import pack_strings
    sch, so = pack_strings([words[int(t)] for t in script])
    return c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads)
