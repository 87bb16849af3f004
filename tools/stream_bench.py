#!/usr/bin/env python3
"""BASELINE configs[4]: a corpus larger than one batch, streamed from pinned host
memory with the upload of batch i+1 overlapping the search of batch i.

  python tools/stream_bench.py [--works 1000000] [--tokens 1000] [--batch 100000]

Prints one JSON line: fanworks/s end to end (PCIe included), the same batches
with uploads and searches serialised, and the pure upload rate."""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from fandom_search_amd import abi, synth, vocab  # noqa: E402
from fandom_search_amd.engine import PinnedBuffer, ScriptIndex, search_stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--works", type=int, default=1_000_000)
    ap.add_argument("--tokens", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=100_000)
    ap.add_argument("--script-tokens", type=int, default=20_000)
    a = ap.parse_args()
    words = synth.vocab_words()
    emb = synth.embedding()
    script = synth.script_tokens(a.script_tokens)
    chars, coff = vocab.pack_strings(words)
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(6),
                     cfg=abi.make_config())
    n_batches = (a.works + a.batch - 1) // a.batch
    rng = np.random.default_rng(5)
    # the corpus lives in pinned host memory (as a tokenising front end would leave it)
    t0 = time.time()
    pinned = PinnedBuffer(a.works * a.tokens, np.uint32)
    tok = pinned.array
    step = 10_000_000
    for lo in range(0, len(tok), step):
        hi = min(len(tok), lo + step)
        tok[lo:hi] = synth._draw(rng, hi - lo, len(words))
    for _ in range(a.works // 50):                    # planted script spans
        ln = int(rng.integers(6, 25))
        src = int(rng.integers(0, len(script) - ln))
        dst = int(rng.integers(0, len(tok) - ln))
        tok[dst:dst + ln] = script[src:src + ln]
    offs = [PinnedBuffer(a.batch + 1, np.uint64) for _ in range(2)]
    print("generated %.2f GB of ids in %.1fs" % (tok.nbytes / 1e9, time.time() - t0), flush=True)

    def batches():
        for b in range(n_batches):
            w0 = b * a.batch
            nb = min(a.batch, a.works - w0)
            o = offs[b & 1].array[:nb + 1]
            o[:] = np.arange(nb + 1, dtype=np.uint64) * np.uint64(a.tokens)
            yield tok[w0 * a.tokens:(w0 + nb) * a.tokens], o

    def run_streamed():
        t = time.perf_counter()
        rows = 0
        for r, st in search_stream(ix, batches(), chars, coff):
            rows += len(r)
        return time.perf_counter() - t, rows

    run_streamed()                                     # warm: allocations, levtab
    dt, rows = run_streamed()

    # serialised: upload, wait, search, one batch at a time
    t = time.perf_counter()
    corpus = None
    up = 0.0
    for tk, o in batches():
        tu = time.perf_counter()
        if corpus is None:
            corpus = ix.corpus(tk, o, chars, coff)
        else:
            corpus.update_begin(tk, o)
            corpus.update_end()
        up += time.perf_counter() - tu
        ix.search(corpus)
    ds = time.perf_counter() - t
    out = {"workload": "%d works x %d tokens in %d batches of %d works, pinned host memory"
                       % (a.works, a.tokens, n_batches, a.batch),
           "streamed_s": dt, "fanworks_per_s_streamed": a.works / dt,
           "serial_s": ds, "fanworks_per_s_serial": a.works / ds,
           "upload_s": up, "upload_GBps": tok.nbytes / up / 1e9,
           "ids_GBps_streamed": tok.nbytes / dt / 1e9, "rows": rows}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
