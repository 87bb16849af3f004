#!/usr/bin/env python3
"""Throughput of the native text encoder by itself (no GPU): a 500-work batch of synthetic files,
fs_textenc_encode_files[_vec] called over and over with 1 .. 32 threads.

  python tools/textenc_bench.py [--works 500] [--tokens 2000] [--dir /dev/shm]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--works", type=int, default=500)
    ap.add_argument("--tokens", type=int, default=2000)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from fandom_search_amd import synth, textenc, vocab
    words = synth.vocab_words()
    script = synth.script_tokens(20000)
    tmp = tempfile.mkdtemp(dir=a.dir)
    try:
        names = synth.write_corpus(os.path.join(tmp, "fan"), 2 * a.works, a.tokens, script, words)
        voc = vocab.Vocab(words, synth.embedding())
        for thr in (1, 2, 4, 8, 16, 32):
            enc = textenc.TextEncoder(voc, threads=thr)
            enc.encode_files(names[:a.works])
            for vec in (False, True):
                ts = []
                for r in range(a.reps):
                    batch = names[a.works:] if r % 2 else names[:a.works]
                    t0 = time.perf_counter()
                    enc._native(batch, {} if vec else None)
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                print(json.dumps({"threads": thr, "vector_ids": vec, "ms_min": round(ts[0] * 1e3, 2),
                                  "ms_median": round(ts[len(ts) // 2] * 1e3, 2),
                                  "works_per_s": round(a.works / ts[len(ts) // 2])}), flush=True)
            enc.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
