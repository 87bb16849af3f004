#!/usr/bin/env python3
"""The mixed-case companion by itself: c2 with 8 % of the fan tokens capitalised (string ids
next to vector ids, search.py:151 vs :166).  Prints ms per search (two in flight); run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FS_LANES", "4")


def main():
    import numpy as np
    import torch
    from fandom_search_amd import abi, synth, vocab
    from fandom_search_amd.engine import ScriptIndex
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    conf = synth.CONFIGS["c2"]
    words, emb = synth.vocab_words(), synth.embedding()
    script = synth.script_tokens(conf["script_tokens"])
    swords = [words[int(t)] for t in script]
    tok, off = synth.corpus_tokens(conf["n_works"], conf["tokens_per_work"], script)
    ix = ScriptIndex(script, swords, emb, synth.lsh_normals(6), cfg=abi.make_config(window_size=6))
    strings = list(words) + [w.capitalize() for w in words]
    schars, scoff = vocab.pack_strings(strings)
    rng = np.random.default_rng(11)
    tok_str = tok.copy()
    sel = rng.random(len(tok_str)) < 0.08
    tok_str[sel] += np.uint32(len(words))
    cs = ix.corpus(tok, off, schars, scoff, tok_str=tok_str)
    rows, st = ix.search(cs)
    cap = len(rows) + 64
    bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda") for _ in range(3)]
    best = None
    for trial in range(3):
        tickets = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            tickets.append(ix.search_begin(cs, bufs[i % 3].data_ptr(), cap, header=True))
            if len(tickets) >= 2:
                ix.search_end(tickets.pop(0))
        while tickets:
            ix.search_end(tickets.pop(0))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        best = ms if best is None else min(best, ms)
    print('{"kernel": "%s", "rows": %d, "ms_per_search": %.4f, "lev_differs": %.3f}'
          % (ix.kernel_name(cs), len(rows), best, float((rows["lev"] != 7).mean())))


if __name__ == "__main__":
    main()
