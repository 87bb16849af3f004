#!/usr/bin/env python3
"""Randomised cross-check of the share rule (the LSH pipeline's prefilter on tables whose vectors
are not unit length: k_share_scan; k_share_gate and the pairs' test in k_lsh_scan) against the key
scan over every window (FS_LSH_SHARE=0; both on the GPU -- the key scan is held against the oracle
by tests/): tables with norms spread by a factor of 1.2 to 50, similarity at three scales,
duplicate and zero rows, long and short vectors (next to the short ones an out-of-vocabulary
name is the heavy slot); scripts with names of their own, the odd ones among them (fewer than
three hot positions); window sizes 3..12, thresholds 0.05..0.25, gamma 0.5..0.85; fan text with
near-synonyms, out-of-vocabulary names, and planted script spans whose *lightest* slots hold
unrelated words (the pairs "at most one slot may differ" misses).

  python tools/stress_share.py [--cases 36] [--seed 1]
"""

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=36)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from fandom_search_amd import abi, synth
    from fandom_search_amd.engine import ScriptIndex
    from fandom_search_amd.vocab import pack_strings

    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        rows = int(rng.choice([1500, 4000]))
        sigma = float(rng.choice([0.05, 0.3, 0.6, 1.0]))
        n = int(rng.choice([3, 4, 5, 6, 6, 6, 7, 8, 9, 10, 11, 12]))
        thr = float(rng.choice([0.05, 0.1, 0.1, 0.25]))
        gamma = float(rng.choice([0.5, 0.7, 0.7, 0.85]))
        oov = float(rng.choice([0.0, 0.08, 0.3]))
        unique = bool(case % 2)
        emb, group = synth.realistic_table(rows=rows, sigma=sigma, seed=11 + case)
        # (norms around 6, 1 or 0.3: next to short table vectors an out-of-vocabulary name -- norm up to
        # sqrt(3) -- is the heavy slot of its window)
        scale = float(rng.choice([1.0, 0.15, 0.05]))
        emb = (emb * np.float32(scale)).astype(np.float32)
        strings, vid = synth.realistic_vector_ids(rows)
        # names whose three hashes fell on fewer than three positions, the same sets under other ids,
        # names that contain them (share_comp's case analysis), as strings of their own
        D, F = 300, abi.FS_OOV_FLAG
        code = lambda a, b, c: np.uint32(F | ((a * D + b) * D + c))
        x, y, z = sorted(((int(vid[2 * rows]) & ~F) // (D * D), ((int(vid[2 * rows]) & ~F) // D) % D, (int(vid[2 * rows]) & ~F) % D))
        family = [code(7, 7, 19), code(7, 19, 19), code(7, 19, 123), code(7, 19, 250), code(55, 55, 55), code(55, 55, 201),
                  code(x, y, y), code(x, x, z), code(x, y, z)]
        strings = list(strings) + ["Odd%d" % i for i in range(len(family))]
        vid = np.concatenate([vid, np.array(family, dtype=np.uint32)])
        odd0 = len(strings) - len(family)
        n_script = int(rng.choice([400, 3000]))
        script = synth._draw(np.random.default_rng(77 + case), n_script, rows)
        n_works, per = int(rng.choice([3, 40])), int(rng.choice([200, 900]))
        tok_str, off = synth.realistic_corpus(n_works, per, script, group, rows, oov_rate=oov, seed=5 + case)
        names = case % 3 != 0                          # two cases in three: a script with names of its own
        script_str = script.astype(np.uint32).copy()
        if names:
            for i in range(0, n_script, int(rng.choice([4, 9]))):
                script_str[i] = (2 * rows + int(rng.integers(0, 25))) if rng.random() < 0.7 else odd0 + int(rng.choice([0, 4, 8]))
            sel = np.nonzero(rng.random(len(tok_str)) < 0.03)[0]      # the odd names in the fan text too
            tok_str[sel] = odd0 + rng.integers(0, len(family), size=len(sel))
        # planted spans: the script's words with the lightest slots of every window-sized piece
        # replaced by unrelated words (and now and then a heavy one)
        norm = np.linalg.norm(emb.astype(np.float64), axis=1)
        for j in range(4 * n_works):
            w = int(rng.integers(0, n_works))
            if per < 3 * n or n_script < 3 * n:
                break
            at = int(off[w]) + int(rng.integers(0, per - 2 * n))
            src = int(rng.integers(0, n_script - 2 * n))
            span = script[src:src + 2 * n].astype(np.uint32).copy()
            order = np.argsort(norm[span])
            keep = script_str[src:src + 2 * n].copy()
            for k in order[:int(rng.integers(0, n))]:
                span[k] = int(rng.integers(0, rows))
            if rng.random() < 0.2:
                span[order[-1]] = int(rng.integers(0, rows))
            if names:                                  # the script's names stay, or become a relative
                for k in range(len(span)):
                    if keep[k] >= 2 * rows and span[k] == script[src + k]:
                        span[k] = keep[k] if keep[k] < odd0 or rng.random() < 0.4 else odd0 + int(rng.integers(0, len(family)))
            tok_str[at:at + len(span)] = span
        tok_vec = vid[tok_str]
        script = vid[script_str]
        swords = [strings[int(t)] for t in script_str]
        chars, coff = pack_strings(strings)
        cfg = abi.make_config(window_size=n, unique_filter=unique, distance_threshold=thr)
        normals = synth.lsh_normals(n)
        os.environ["FS_SHARE_GAMMA"] = str(gamma)
        results = []
        for share in ("0", "35", "3", "43"):
            os.environ["FS_LSH_SHARE"] = share
            ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
            c = ix.corpus(tok_vec, off, chars, coff, tok_str=tok_str)
            r, st = ix.search(c)
            results.append((r.tobytes(), int(st.matches), ix.kernel_name(c), ix.share_info()["flags"], int(st.candidates)))
            ix.close()
        ok = all(x[:2] == results[0][:2] for x in results)
        r0 = np.frombuffer(results[0][0], dtype=abi.ROW_DTYPE)
        print("case %3d rows=%4d sigma=%.2f scale=%.2f names=%d n=%d thr=%.2f gamma=%.2f oov=%.2f u%d script=%4d tokens=%6d records=%5d inexact=%5d "
              "kernels=%s flags=%s candidates=%s %s"
              % (case, rows, sigma, scale, names, n, thr, gamma, oov, unique, n_script, len(tok_vec), len(r0),
                 int((np.abs(r0["dist"]) > 1e-9).sum()), [x[2] for x in results], [x[3] for x in results],
                 [x[4] for x in results], "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("mismatches: %d" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
