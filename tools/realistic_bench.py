#!/usr/bin/env python3
"""The search on synth.realistic_table (unnormalised vectors, three scales of similarity,
duplicate and zero rows) under fan text with and without out-of-vocabulary names: which
pipeline runs, the component statistics of the near-synonym prefilter, fanworks/s.

  python tools/realistic_bench.py [--works 2000] [--rows 20000] [--oov 0.08]
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(works=2000, rows=20000, oov=0.08, tokens=2000, script_tokens=20000, window=6, reps=3, companion=None, counts=True):
    """companion: called with (index, corpus, works, tokens in the corpus) before they are closed; the dict it
    returns is merged into the result (bench.py measures the search with others in flight through it)."""
    from fandom_search_amd import abi, synth
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings
    emb, group = synth.realistic_table(rows=rows)
    strings, vid = synth.realistic_vector_ids(rows)
    script = synth._draw(np.random.default_rng(77), script_tokens, rows)
    tok_str, off = synth.realistic_corpus(works, tokens, script, group, rows, oov_rate=oov)
    tok_vec = vid[tok_str]
    chars, coff = pack_strings(strings)
    swords = [strings[int(t)] for t in script]
    t0 = time.perf_counter()
    ix = ScriptIndex(script, swords, emb, synth.lsh_normals(window), cfg=abi.make_config(window_size=window))
    t_index = time.perf_counter() - t0
    c = ix.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
    rows_out, st = ix.search(c)
    best = None
    for _ in range(reps):
        rows_out, st = ix.search(c, reuse=True)
        if st.total_ms > 0:
            best = st.total_ms if best is None else min(best, st.total_ms)
    out = {"rows": rows, "works": works, "tokens_per_work": tokens, "oov_rate": oov, "window": window,
           "kernel": ix.kernel_name(c), "path": "lsh" if st.path == abi.FS_MODE_GENERAL else "exact",
           "c_max": ix.info["c_max"], "norm_min": ix.info["norm_min"], "norm_max": ix.info["norm_max"],
           "index_s": round(t_index, 2), "ms_per_search_alone": best,
           "value_alone": works / (best * 1e-3) if best else None, "unit": "fanworks/s",
           "windows_per_s": st.windows_processed / (best * 1e-3) if best else None,
           "candidates": int(st.candidates), "lsh_pending": int(st.lsh_pending),
           "records": int(len(rows_out)), "inexact_records": int((np.abs(rows_out["dist"]) > 1e-9).sum()),
           "oov_share_of_tokens": float((tok_vec & abi.FS_OOV_FLAG != 0).mean())}
    import torch
    cap = len(rows_out) + 64
    buf = torch.zeros(cap * 32, dtype=torch.uint8, device="cuda")
    prof = None
    for _ in range(3):
        prof = ix.profile(c, buf.data_ptr(), cap)
    out["kernels_us"] = [[k, round(ms * 1e3, 1)] for k, ms in prof]
    out["share_rule"] = ix.share_info()
    if out["share_rule"]["flags"] & 32 and counts:
        # what passes what, from an index of its own that counts (the counters cost a few atomic
        # additions per sub-tile: not on the timed index)
        os.environ["FS_SHARE_COUNT"] = "1"
        ixc = ScriptIndex(script, swords, emb, synth.lsh_normals(window), cfg=abi.make_config(window_size=window))
        del os.environ["FS_SHARE_COUNT"]
        cc = ixc.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
        ixc.search(cc)
        ixc.share_counts()
        ixc.search(cc, reuse=True)
        n = ixc.share_counts()
        w = max(n["windows"], 1)
        out["share_rule"]["one_search"] = n
        out["share_rule"]["shares"] = {
            "windows_with_a_key_in_the_filter": n["windows_with_a_key_in_the_filter"] / w,
            "windows_flagged": n["windows_flagged"] / w,
            "windows_flagged_as_they_are (fallback)": n["windows_flagged_as_they_are"] / w,
            "pairs_tested_per_window": n["pairs_tested"] / w, "distances_per_window": n["distances"] / w}
        cc.close()
        ixc.close()
    sizes, used = ix.component_sizes()
    if not len(sizes):
        out["components"] = {"count": 0, "in_use": False,
                             "note": "the graph of near pairs was not kept: more than 2^23 pairs of (script vector, "
                                     "table vector) lie above the line a pair must be above to be near -- with norms "
                                     "spread over a factor of ten the line is below cosine -1 for typical pairs"}
    if len(sizes):
        edges = [1, 2, 3, 5, 9, 17, 65, 257, 1025, 1 << 30]
        out["components"] = {"count": int(len(sizes)), "largest": int(sizes.max()), "in_use": used,
                             "relation": "cosine > %.2f (the share rule)" % out["share_rule"]["gamma"]
                                         if out["share_rule"]["flags"] else "near pairs of section 4",
                             "histogram": {"%d-%d" % (lo, hi - 1) if hi < (1 << 30) else "%d+" % lo:
                                           int(((sizes >= lo) & (sizes < hi)).sum())
                                           for lo, hi in zip(edges[:-1], edges[1:])},
                             "share_of_table_in_largest": float(sizes.max()) / rows}
    if companion is not None:
        out.update(companion(ix, c, works, len(tok_vec)))
    c.close()
    ix.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--works", type=int, default=2000)
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--oov", type=float, default=0.08)
    ap.add_argument("--window", type=int, default=6)
    ap.add_argument("--no-counts", action="store_true", help="no second index that counts what passes what (under a profiler)")
    a = ap.parse_args()
    for oov in (a.oov, 0.0):
        print(json.dumps(run(a.works, a.rows, oov, window=a.window, counts=not a.no_counts)), flush=True)
