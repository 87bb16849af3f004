#!/usr/bin/env python3
"""Sweep scan-kernel variants on the GPU (diagnostic, not part of the product).

  python tools/scan_sweep.py [--workload c2|c3shard|c3] [--window 6]

Variants are selected through the FS_SCAN_* / FS_FILTER_* environment switches
of fs_scan.hip / fs_api.hip; every variant is timed as back-to-back launches of
the scan kernel alone (fs_scan_benchmark)."""

import argparse
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from fandom_search_amd import abi, synth, vocab  # noqa: E402
from fandom_search_amd.engine import ScriptIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--window", type=int, default=6)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--works", type=int, default=0)
    ap.add_argument("--filters", default="14")
    ap.add_argument("--unrolls", default="1,2,4,8")
    ap.add_argument("--halos", default="loads,shuffle")
    ap.add_argument("--blocks", default="0")
    ap.add_argument("--flags", default="word,direct")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    conf = dict(synth.CONFIGS[a.workload])
    if a.works:
        conf["n_works"] = a.works
    words = synth.vocab_words()
    emb = synth.embedding()
    normals = synth.lsh_normals(a.window)
    script = synth.script_tokens(conf["script_tokens"])
    swords = [words[int(t)] for t in script]
    chars, coff = vocab.pack_strings(words)
    t0 = time.time()
    rng = np.random.default_rng(1)
    # the scan's speed does not depend on the planted spans: draw fast
    n_tok = conf["n_works"] * conf["tokens_per_work"]
    if n_tok > 50_000_000:
        tok = synth._draw(rng, n_tok, len(words))
        off = np.arange(conf["n_works"] + 1, dtype=np.uint64) * np.uint64(conf["tokens_per_work"])
    else:
        tok, off = synth.corpus_tokens(conf["n_works"], conf["tokens_per_work"], script)
    print("corpus %d tokens in %.1fs" % (len(tok), time.time() - t0), flush=True)
    out = []
    for lw in [int(x) for x in a.filters.split(",")]:
        os.environ["FS_FILTER_LOG2_WORDS"] = str(lw)
        ix = ScriptIndex(script, swords, emb, normals, cfg=abi.make_config(window_size=a.window))
        corpus = ix.corpus(tok, off, chars, coff)
        variants = list(itertools.product(a.halos.split(","),
                                          [int(x) for x in a.unrolls.split(",")],
                                          [int(x) for x in a.blocks.split(",")],
                                          a.flags.split(",")))
        best = {}
        for _round in range(a.rounds):           # interleaved rounds in one process
            for v in variants:
                halo, un, blk, fl = v
                os.environ["FS_SCAN_HALO"] = halo
                os.environ["FS_SCAN_UNROLL"] = str(un)
                os.environ["FS_SCAN_BLOCKS_PER_CU"] = str(blk)
                os.environ["FS_SCAN_FLAGS"] = fl
                ix.reload_switches()
                ms = ix.scan_benchmark(corpus, a.reps)
                best.setdefault(v, []).append(ms)
        for v in variants:
            halo, un, blk, fl = v
            ms = min(best[v])
            med = sorted(best[v])[len(best[v]) // 2]
            gbs = 4.0 * len(tok) / (ms * 1e-3) / 1e9
            rec = dict(filter_kb=(4 << lw) // 1024, halo=halo, unroll=un, blocks_per_cu=blk,
                       flags=fl, ms=round(ms, 5), ms_median=round(med, 5), GBps=round(gbs, 1),
                       frac=round(gbs / 8000, 4))
            out.append(rec)
            print(json.dumps(rec), flush=True)
        corpus.close()
        ix.close()
    best = max(out, key=lambda r: r["GBps"])
    print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
