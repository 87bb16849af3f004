#!/usr/bin/env python3
"""Synchronous fs_search_corpus with host rows on four C2 batches in rotation: wall time per
search against the GPU time inside it (fs_stats.total_ms).  FS_HOST_ZEROCOPY=0 / FS_LANES=4 as
environment switches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fandom_search_amd import abi, synth, vocab
from fandom_search_amd.engine import ScriptIndex
conf = synth.CONFIGS["c2"]
words, emb = synth.vocab_words(), synth.embedding()
script = synth.script_tokens(conf["script_tokens"])
swords = [words[int(t)] for t in script]
chars, coff = vocab.pack_strings(words)
ix = ScriptIndex(script, swords, emb, synth.lsh_normals(6), cfg=abi.make_config(window_size=6))
cs = []
for r in range(4):
    t, o = synth.corpus_tokens(conf["n_works"], conf["tokens_per_work"], script, first_work=r * conf["n_works"])
    cs.append(ix.corpus(t, o, chars, coff))
for i in range(8):
    ix.search(cs[i % 4], reuse=True)
N = 60
tot = []
t0 = time.perf_counter()
for i in range(N):
    rows, st = ix.search(cs[i % 4], reuse=True)
    tot.append(st.total_ms)
wall = (time.perf_counter() - t0) / N * 1e3
print("wall ms/search %.4f  gpu total_ms mean %.4f  scan_ms %.4f rows %d" % (wall, float(np.mean(tot)), st.scan_ms, len(rows)))
