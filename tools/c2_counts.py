#!/usr/bin/env python3
"""Candidates, matches and records of one configs[1] batch (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fandom_search_amd import abi, synth, vocab
from fandom_search_amd.engine import ScriptIndex
conf = synth.CONFIGS["c2"]
words, emb = synth.vocab_words(), synth.embedding()
script = synth.script_tokens(conf["script_tokens"])
tok, off = synth.corpus_tokens(conf["n_works"], conf["tokens_per_work"], script)
chars, coff = vocab.pack_strings(words)
for n in (6, 4):
    ix = ScriptIndex(script, [words[int(t)] for t in script], emb, synth.lsh_normals(n), cfg=abi.make_config(window_size=n))
    c = ix.corpus(tok, off, chars, coff)
    rows, st = ix.search(c)
    print(n, ix.kernel_name(c), "tokens", len(tok), "candidates", st.candidates, "matches", st.matches, "rows", len(rows))
