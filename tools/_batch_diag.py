
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fandom_search_amd import abi, synth, vocab
from fandom_search_amd.engine import ScriptIndex
words = synth.vocab_words(); script = synth.script_tokens(20000)
tok, off = synth.corpus_tokens_parallel(10000, 2000, script)
emb, perm = synth.clustered_table(); tok = synth.synonym_swaps(tok, perm)
chars, coff = vocab.pack_strings(words); swords = [words[int(t)] for t in script]
cfg = abi.make_config(window_size=6); normals = synth.lsh_normals(6)
for diag in (0x200000, 0x400000):
    os.environ["FS_LSH_DIAG"] = str(diag)
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    c = ix.corpus(tok, off, chars, coff)
    rows, st = ix.search(c)
    buf = torch.zeros(32 + (len(rows) + 64) * 32 + 40000000, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        prof = ix.profile(c, buf.data_ptr() + 32, len(rows) + 64 + 1000000)
    print(hex(diag), [(k, round(ms * 1e3, 1)) for k, ms in prof], "rows", len(rows), "pending", st.lsh_pending, flush=True)
    ix.close()
