#!/usr/bin/env python3
"""A/B timing of whole search steps under different FS_* switches, one process.

  python tools/step_bench.py [--workload c2] [--steps 300] [--rounds 3] \
      "FS_SCAN_ROWS=1" "FS_SCAN_ROWS=0" "FS_SCAN_ROWS=1 FS_LANES=4"

Each variant gets its own index (the switches are read at fs_index_create) over the
same resident corpus; the variants are timed in interleaved rounds (two searches in
flight, rows left in HBM, as bench.py does) and the best and median round are printed
with the scan kernel's own duration (HIP events on every 4th search).
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
    ap.add_argument("variants", nargs="*", default=["FS_SCAN_ROWS=1", "FS_SCAN_ROWS=0",
                                                    "FS_SCAN_ROWS=1 FS_LANES=4"])
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--works", type=int, default=0)
    ap.add_argument("--window", type=int, default=6)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=2)
    ap.add_argument("--rotate", type=int, default=1,
                    help="distinct batches per variant (4: 320 MB of c2 ids, more than the Infinity Cache: from HBM)")
    a = ap.parse_args()

    import torch
    from fandom_search_amd import abi, synth, vocab
    from fandom_search_amd.engine import ScriptIndex

    conf = dict(synth.CONFIGS[a.workload])
    if a.works:
        conf["n_works"] = a.works
    words, emb = synth.vocab_words(), synth.embedding()
    normals = synth.lsh_normals(a.window)
    script = synth.script_tokens(conf["script_tokens"])
    swords = [words[int(t)] for t in script]
    chars, coff = vocab.pack_strings(words)
    batches = [synth.corpus_tokens_parallel(conf["n_works"], conf["tokens_per_work"], script,
                                            first_work=r * conf["n_works"]) for r in range(max(1, a.rotate))]
    tok, off = batches[0]
    cfg = abi.make_config(window_size=a.window)

    setups = []
    for v in a.variants:
        env = dict(kv.split("=", 1) for kv in v.split())
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
        for k, o in old.items():
            if o is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = o
        corpora = [ix.corpus(t, o, chars, coff) for t, o in batches]
        corpus = corpora[0]
        rows, st = ix.search(corpus)
        for cc in corpora[1:]:
            cap_r, _ = ix.search(cc)
            rows = rows if len(rows) >= len(cap_r) else cap_r
        cap = len(rows) + 64
        bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda")
                for _ in range(a.inflight + 1)]
        ix.set_scan_timing(4)
        rows, st = ix.search(corpus)
        setups.append(dict(name=v, ix=ix, corpus=corpus, corpora=corpora, bufs=bufs, cap=cap, n_rows=len(rows),
                           crc=hash(rows.tobytes()), ms=[], scan=[]))
    if not any("DIAG" in s["name"] for s in setups):
        assert len({s["crc"] for s in setups}) == 1, "variants disagree on the rows"

    def run(s, steps):
        ix, corpus, bufs, cap = s["ix"], s["corpus"], s["bufs"], s["cap"]
        tickets, scan = [], []
        for i in range(steps):
            tickets.append(ix.search_begin(s["corpora"][i % len(s["corpora"])], bufs[i % len(bufs)].data_ptr(), cap, header=True))
            if len(tickets) >= a.inflight:
                n, st = ix.search_end(tickets.pop(0))
                s["last"] = (int(st.candidates), int(st.lsh_pending))
                if st.scan_ms > 0:
                    scan.append(st.scan_ms)
        while tickets:
            ix.search_end(tickets.pop(0))
        return scan

    for s in setups:
        run(s, 20)
    for _ in range(a.rounds):
        for s in setups:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            scan = run(s, a.steps)
            torch.cuda.synchronize()
            s["ms"].append((time.perf_counter() - t0) / a.steps * 1e3)
            s["scan"].append(float(np.mean(scan)) if scan else 0.0)
    n_tok = len(tok)
    for s in setups:
        best = min(s["ms"])
        print(json.dumps(dict(variant=s["name"], step_us=round(best * 1e3, 2),
                              step_us_median=round(sorted(s["ms"])[len(s["ms"]) // 2] * 1e3, 2),
                              scan_kernel_us=round(min(s["scan"]) * 1e3, 2), rows=s["n_rows"],
                              candidates=s.get("last", (0, 0))[0], lsh_pending=s.get("last", (0, 0))[1],
                              step_frac_of_8TBs=round((4.0 * n_tok + 32.0 * s["n_rows"]) / (best * 1e-3) / 8e12, 4))),
              flush=True)


if __name__ == "__main__":
    main()
