#!/usr/bin/env python3
"""A/B of FS_* switches on ONE index over the SAME device corpora.

  FS_LANES=1 python tools/diag_ab.py [--rotate 4] [--steps 200] [--rounds 6] "FS_DIAG=0" "FS_DIAG=64" ...

tools/step_bench.py gives every variant an index and corpora of its own; on a C2 batch the
placement of those buffers alone moves the search kernel by +-0.7 us (the same variant at two
positions of one run: 33.0 / 32.7, 31.8 / 33.2 us), which is what a switch of the prologue is worth.
Here the switches are re-read on a live index (fs_index_reload_switches) between interleaved
rounds, so the variants differ in nothing but the switch.  Only switches that are read per
search can be compared this way (FS_DIAG bits of the kernels, not FS_LANES or table sizes).
Prints per variant the search kernel's dispatch-to-completion time (HIP events, every search)
and the step time, mean and best over the rounds.
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
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--window", type=int, default=6)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--inflight", type=int, default=2)
    ap.add_argument("--rotate", type=int, default=4)
    a = ap.parse_args()
    os.environ.setdefault("FS_LANES", "1")

    import torch
    from fandom_search_amd import abi, synth, vocab
    from fandom_search_amd.engine import ScriptIndex

    conf = dict(synth.CONFIGS[a.workload])
    words, emb = synth.vocab_words(), synth.embedding()
    script = synth.script_tokens(conf["script_tokens"])
    swords = [words[int(t)] for t in script]
    chars, coff = vocab.pack_strings(words)
    ix = ScriptIndex(script, swords, emb, synth.lsh_normals(a.window), cfg=abi.make_config(window_size=a.window))
    corpora = []
    cap = 0
    for r in range(max(1, a.rotate)):
        t, o = synth.corpus_tokens_parallel(conf["n_works"], conf["tokens_per_work"], script,
                                            first_work=r * conf["n_works"])
        corpora.append(ix.corpus(t, o, chars, coff))
        rows, _ = ix.search(corpora[-1])
        cap = max(cap, len(rows) + 64)
    bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda") for _ in range(a.inflight + 1)]
    ix.set_scan_timing(1)

    def set_env(v):
        env = dict(kv.split("=", 1) for kv in v.split())
        for k, val in env.items():
            os.environ[k] = val
        ix.reload_switches()
        return env

    def run(steps):
        tickets, scan, rows = [], [], 0
        for i in range(steps):
            tickets.append(ix.search_begin(corpora[i % len(corpora)], bufs[i % len(bufs)].data_ptr(), cap, header=True))
            if len(tickets) >= a.inflight:
                n, st = ix.search_end(tickets.pop(0))
                rows = n
                if st.scan_ms > 0:
                    scan.append(st.scan_ms)
        while tickets:
            n, st = ix.search_end(tickets.pop(0))
            if st.scan_ms > 0:
                scan.append(st.scan_ms)
        return scan, rows

    res = {v: dict(kernel=[], step=[], rows=None) for v in a.variants}
    for v in a.variants:
        set_env(v)
        run(20)
    for _ in range(a.rounds):
        for v in a.variants:
            set_env(v)
            run(8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            scan, rows = run(a.steps)
            torch.cuda.synchronize()
            res[v]["step"].append((time.perf_counter() - t0) / a.steps * 1e6)
            res[v]["kernel"].append(float(np.mean(scan)) * 1e3)
            res[v]["rows"] = int(rows)
    for v in a.variants:
        r = res[v]
        print(json.dumps(dict(variant=v, kernel_us_mean=round(float(np.mean(r["kernel"])), 2),
                              kernel_us_min=round(min(r["kernel"]), 2), kernel_us_max=round(max(r["kernel"]), 2),
                              step_us_mean=round(float(np.mean(r["step"])), 2), step_us_min=round(min(r["step"]), 2),
                              rows=r["rows"])), flush=True)


if __name__ == "__main__":
    main()
