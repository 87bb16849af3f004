#!/usr/bin/env python3
"""Randomised cross-check of the LSH pipeline's integer prefilters and shortcuts (3-gram
prefilter, wildcard keys, identical-id shortcuts, per-window Levenshtein table, per-n-gram
records, exact one-slot map) against the same pipeline with all of them switched off (both on the GPU; the unfiltered pipeline is held
against the oracle by tests/): window sizes 8..12 on the synthetic table, planted spans with
zero, one or two substituted tokens, script words spelled differently from the table.

  python tools/stress_lsh.py [--cases 40] [--seed 1]
"""

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCHES = ("FS_LSH_PREFILTER", "FS_LSH_WILD", "FS_LSH_SELFLEV", "FS_LSH_GRAMTAB", "FS_LSH_WMAP", "FS_LSH_SYN",
            "FS_NEAR_FUSED", "FS_LSH_BATCH", "FS_LSH_EMAP")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from fandom_search_amd import abi, synth
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings

    rng = np.random.default_rng(a.seed)
    words, emb_u = synth.vocab_words(), synth.embedding()
    emb_c, perm = synth.clustered_table()
    chars, coff = pack_strings(words)
    bad = 0
    for case in range(a.cases):
        # round 5: every other case on the table with near-synonyms (component-id prefilters,
        # n = 6 .. 10 there), 10 % of the fan tokens swapped for a synonym; both settings of the
        # UniqueFilter; and the pending windows through k_lsh_pkeys / k_lsh_enum / k_lsh_batch
        # however few they are (FS_LSH_DEFER_MIN=0) on every third case
        clustered = case % 2 == 1
        emb = emb_c if clustered else emb_u
        n = int(rng.choice([6, 7, 8, 10])) if clustered else int(rng.choice([8, 9, 10, 12]))
        unique = bool(case % 4 >= 2)
        os.environ["FS_LSH_DEFER_MIN"] = "0" if case % 3 == 0 else "8192"
        n_script = int(rng.choice([300, 2000, 8000]))
        script = synth.script_tokens(n_script)
        swords = [words[int(t)].upper() if rng.random() < 0.1 else words[int(t)] for t in script]
        n_works = int(rng.choice([1, 5, 30]))
        lengths = rng.integers(0, int(rng.choice([30, 800, 6000])) + 1, size=n_works)
        parts = []
        for i, L in enumerate(lengths):
            t = synth.fanwork_tokens(case * 100 + i, int(L), script).copy() if L else np.zeros(0, np.uint32)
            for _ in range(int(rng.integers(0, 6))):           # spans with 0, 1 or 2 odd tokens
                if L > 2 * n and n_script > n + 2:
                    at = int(rng.integers(0, L - n)); src = int(rng.integers(0, n_script - n))
                    t[at:at + n] = script[src:src + n]
                    for _ in range(int(rng.integers(0, 3))):
                        t[at + int(rng.integers(0, n))] = int(rng.integers(0, len(words)))
            parts.append(t)
        off = np.zeros(n_works + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in parts])
        tok = np.concatenate(parts).astype(np.uint32) if n_works else np.zeros(0, np.uint32)
        if clustered and len(tok):
            tok = synth.synonym_swaps(tok, perm, seed=case)
        cfg = abi.make_config(window_size=n, unique_filter=unique)
        normals = synth.lsh_normals(n)
        results = []
        for on in ("1", "0"):
            for k in SWITCHES:
                os.environ[k] = on
            ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
            c = ix.corpus(tok, off, chars, coff)
            rows, st = ix.search(c)
            results.append((rows.tobytes(), int(st.matches), int(st.windows_processed), ix.kernel_name(c),
                            int(ix.info["path"])))
            ix.close()
        ok = results[0][:3] == results[1][:3]
        print("case %3d %s u%d n=%2d script=%5d works=%3d tokens=%7d rows=%6d inexact=%4d %s/%s %s"
              % (case, "clustered" if clustered else "synthetic", unique, n, n_script, n_works, len(tok), len(results[0][0]) // 32,
                 int((np.frombuffer(results[0][0], dtype=abi.ROW_DTYPE)["dist"] > 1e-9).sum()),
                 results[0][3], results[1][3], "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("mismatches: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
