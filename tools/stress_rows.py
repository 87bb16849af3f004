#!/usr/bin/env python3
"""Randomised cross-check of k_scan_rows against the chained kernels it replaces (both on
the GPU; the chain is held against the oracle by tests/): random batch sizes from a few
tokens to a few million, window sizes 2..8, planted-quote densities, one or four lanes,
all three record formats.

  python tools/stress_rows.py [--cases 60] [--seed 1]
"""

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--strings", action="store_true",
                    help="batches with string ids of their own (a random share of the tokens capitalised): "
                         "k_scan_rows' per-hit Levenshtein form against the chained wave-per-pair kernels")
    a = ap.parse_args()
    import torch
    from fandom_search_amd import abi, synth
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings

    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        n = int(rng.integers(2, 9))
        V = 256 if n >= 7 else synth.VOCAB_SIZE          # n = 7, 8: a table the proof accepts
        if n >= 7:
            words = synth.vocab_words(V)
            emb = np.eye(V, synth.EMB_DIM, dtype=np.float32)
        else:
            words, emb = synth.vocab_words(), synth.embedding()
        chars, coff = pack_strings(words)
        n_script = int(rng.choice([50, 700, 5000, 20000]))
        script = synth.script_tokens(n_script, vocab_size=V)
        n_works = int(rng.choice([1, 2, 7, 40, 300]))
        max_len = int(rng.choice([3, 40, 600, 5000, 20000]))
        lengths = rng.integers(0, max_len + 1, size=n_works)
        parts = []
        for i, L in enumerate(lengths):
            t = synth.fanwork_tokens(case * 1000 + i, int(L), script, V) if L else np.zeros(0, np.uint32)
            # dense quotes now and then: every window a hit
            if L > 3 * n and rng.random() < 0.3 and n_script > L:
                q0 = int(rng.integers(0, n_script - L // 2))
                t = t.copy()
                t[: L // 2] = script[q0:q0 + L // 2]
            parts.append(t)
        off = np.zeros(n_works + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in parts])
        tok = np.concatenate(parts).astype(np.uint32) if n_works else np.zeros(0, np.uint32)
        lanes = int(rng.choice([1, 4]))
        caprow = int(rng.choice([0, 0, 2, 16]))
        tok_str = None
        if a.strings:
            strings = list(words) + [w.capitalize() for w in words]
            chars, coff = pack_strings(strings)
            share = float(rng.choice([0.0, 0.02, 0.2, 0.9]))
            tok_str = tok.copy()
            tok_str[rng.random(len(tok)) < share] += np.uint32(len(words))
        cfg = abi.make_config(window_size=n)
        normals = synth.lsh_normals(n)
        results = []
        for rows_kernel in (1, 0):
            os.environ["FS_SCAN_ROWS"] = str(rows_kernel)
            if a.strings:
                # 1: k_scan_rows with the per-hit Levenshtein in the hit's lane and the table's
                # records; 0: the chained kernels, a wave per (hit, rank) pair, no table
                os.environ["FS_STR_LEVTAB"] = str(rows_kernel)
                os.environ["FS_STR_FAST"] = str(rows_kernel)
            os.environ["FS_LANES"] = str(lanes)
            if caprow and rows_kernel:
                os.environ["FS_RANGES_CAPROW"] = str(caprow)
            else:
                os.environ.pop("FS_RANGES_CAPROW", None)
            ix = ScriptIndex(script, [words[int(t)] for t in script], emb, normals, cfg=cfg)
            c = ix.corpus(tok, off, chars, coff, tok_str=tok_str)
            rows, st = ix.search(c)
            name = ix.kernel_name(c)
            results.append((rows.tobytes(), int(st.matches), int(st.windows_processed), name))
            ix.close()
        ok = results[0][:3] == results[1][:3]
        print("case %3d n=%d script=%5d works=%3d tokens=%8d lanes=%d caprow=%2d rows=%7d %s %s"
              % (case, n, n_script, n_works, len(tok), lanes, caprow, len(results[0][0]) // 32,
                 results[0][3], "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("mismatches: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
