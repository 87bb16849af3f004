#!/usr/bin/env python3
"""Host-side cost of queueing a search (diagnostic, not part of the product).

  [FS_LANES=n] python tools/host_time.py

C2 corpus resident in HBM; 200 steps of fs_search_corpus_begin (queue the next
search) + fs_search_corpus_end (collect the previous one).  Prints the wall time
per step and how it splits into queueing (kernel launches, event records) and
waiting for the GPU.  Measured: 32-35 us of queueing per search against 66 us
of GPU work, so with two searches in flight the host is not the bottleneck."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from fandom_search_amd import abi, synth, vocab
from fandom_search_amd.engine import ScriptIndex
words = synth.vocab_words(); emb = synth.embedding(); normals = synth.lsh_normals(6)
script = synth.script_tokens(20000)
chars, coff = vocab.pack_strings(words)
tok, off = synth.corpus_tokens(10000, 2000, script)
ix = ScriptIndex(script, [words[int(t)] for t in script], emb, normals, cfg=abi.make_config())
c = ix.corpus(tok, off, chars, coff)
cap = 400000
bufs = [torch.zeros(cap * 32, dtype=torch.uint8, device="cuda") for _ in range(4)]
for b in bufs: ix.search_device(c, b.data_ptr(), cap)
tb = te = 0.0; N = 200
t_all = time.perf_counter()
prev = None
for i in range(N):
    t0 = time.perf_counter()
    t = ix.search_begin(c, bufs[i % 4].data_ptr(), cap)
    t1 = time.perf_counter()
    tb += t1 - t0
    if prev is not None:
        ix.search_end(prev)
        te += time.perf_counter() - t1
    prev = t
ix.search_end(prev)
tot = time.perf_counter() - t_all
print("per step %.1f us; begin %.1f us; end(wait) %.1f us" % (tot / N * 1e6, tb / N * 1e6, te / N * 1e6))
