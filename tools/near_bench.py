#!/usr/bin/env python3
"""The LSH pipeline behind its integer prefilters, variant against variant in one process:
per-kernel times of one search (fs_search_profile), the whole search alone, and the step
with four searches in flight (as bench.py measures the headline).

  python tools/near_bench.py --window 8 [--table synthetic|clustered] [--works 10000] \
      "FS_NEAR_FUSED=1" "FS_NEAR_FUSED=0"
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*", default=["FS_NEAR_FUSED=1", "FS_NEAR_FUSED=0"])
    ap.add_argument("--window", type=int, default=8)
    ap.add_argument("--table", default="synthetic")
    ap.add_argument("--works", type=int, default=10000)
    ap.add_argument("--tokens", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--unique", type=int, default=-1)
    a = ap.parse_args()
    os.environ["FS_LANES"] = str(a.lanes)
    import torch
    from fandom_search_amd import abi, synth, vocab
    from fandom_search_amd.engine import ScriptIndex

    words = synth.vocab_words()
    script = synth.script_tokens(20000)
    tok, off = synth.corpus_tokens_parallel(a.works, a.tokens, script)
    if a.table == "clustered":
        emb, perm = synth.clustered_table()
        tok = synth.synonym_swaps(tok, perm)
    else:
        emb = synth.embedding()
    chars, coff = vocab.pack_strings(words)
    swords = [words[int(t)] for t in script]
    kw = {} if a.unique < 0 else {"unique_filter": bool(a.unique)}
    cfg = abi.make_config(window_size=a.window, **kw)
    normals = synth.lsh_normals(a.window)
    idx = []
    for v in a.variants:
        env = dict(kv.split("=", 1) for kv in v.split())
        for k, val in env.items():
            os.environ[k] = val
        ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
        c = ix.corpus(tok, off, chars, coff)
        for k in env:
            del os.environ[k]
        rows, st = ix.search(c)
        idx.append((v, ix, c, rows.tobytes(), st))
    ref = idx[0][3]
    cap = len(idx[0][3]) // 32 + 64
    bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda") for _ in range(a.lanes + 1)]
    res = {v: {"step_ms": [], "alone_ms": []} for v, *_ in idx}
    for v, ix, c, b, st in idx:
        assert b == ref, "records differ between %s and %s" % (v, idx[0][0])
        for _ in range(2 * a.lanes):
            ix.search_end(ix.search_begin(c, bufs[0].data_ptr(), cap, header=True))
    for _ in range(a.rounds):
        for v, ix, c, b, st in idx:
            ix.set_scan_timing(1 << 20)
            tickets = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                tickets.append(ix.search_begin(c, bufs[i % len(bufs)].data_ptr(), cap, header=True))
                if len(tickets) >= a.lanes:
                    ix.search_end(tickets.pop(0))
            while tickets:
                ix.search_end(tickets.pop(0))
            torch.cuda.synchronize()
            res[v]["step_ms"].append((time.perf_counter() - t0) / a.steps * 1e3)
            ix.set_scan_timing(1)
            best = None
            for _ in range(3):
                _, s2 = ix.search(c, reuse=True)
                if s2.total_ms > 0:
                    best = s2.total_ms if best is None else min(best, s2.total_ms)
            res[v]["alone_ms"].append(best)
    for v, ix, c, b, st in idx:
        prof = None
        for _ in range(3):
            prof = ix.profile(c, bufs[0].data_ptr() + 32, cap)
        r = res[v]
        print(json.dumps({"variant": v, "window": a.window, "table": a.table, "works": a.works,
                          "kernel": ix.kernel_name(c),
                          "step_ms_best": round(min(r["step_ms"]), 4), "step_ms_median": round(float(np.median(r["step_ms"])), 4),
                          "alone_ms": round(min(x for x in r["alone_ms"] if x), 4),
                          "kernels_us": [[k, round(ms * 1e3, 1)] for k, ms in prof],
                          "candidates": int(st.candidates), "lsh_pending": int(st.lsh_pending),
                          "matches": int(st.matches), "rows": len(b) // 32}), flush=True)


if __name__ == "__main__":
    main()
