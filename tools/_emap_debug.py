import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from fandom_search_amd import abi, synth
from fandom_search_amd.engine import ScriptIndex
from fandom_search_amd.vocab import pack_strings
from tests import util
n, unique = 9, 1
words = synth.vocab_words(); emb = synth.embedding()
chars, coff = pack_strings(words)
script = synth.script_tokens(4000)
tok, off = util.ragged_corpus([800] * 14 + [0, n - 1, n, 1700], script)
tok = tok.copy()
cos = emb[script[:600]] @ emb.T
cos[np.arange(600), script[:600]] = -1.0
best = cos.argmax(axis=1)
for j in range(n):
    src = 300 + 20 * j
    at = int(off[j % 14]) + 100 + 40 * j
    tok[at:at + n] = script[src:src + n]
    tok[at + j] = best[src + j] if j % 2 == 0 else (int(tok[at + j]) + 17) % len(words)
for j in range(3):
    at = int(off[10 + j]) + 600
    tok[at:at + n] = script[900 + 30 * j:900 + 30 * j + n]
    tok[at + 1] = (int(tok[at + 1]) + 5) % len(words)
    tok[at + n - 2] = (int(tok[at + n - 2]) + 9) % len(words)
cfg = abi.make_config(window_size=n, unique_filter=unique)
normals = synth.lsh_normals(n)
swords = [words[int(t)].upper() if i % 7 == 0 else words[int(t)] for i, t in enumerate(script)]
res = {}
for name, env in (("emap", {"FS_LSH_WMAP": "0"}), ("walk", {"FS_LSH_WMAP": "0", "FS_LSH_EMAP": "0"})):
    for k, v in env.items(): os.environ[k] = v
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    c = ix.corpus(tok, off, chars, coff)
    rows, st = ix.search(c)
    res[name] = rows.copy()
    print(name, len(rows), st.matches, st.lsh_pending, flush=True)
    for k in env: del os.environ[k]
    ix.close()
a, b = res["emap"], res["walk"]
ka = {(int(r["work"]), int(r["fan_ix"])): r for r in a}
kb = {(int(r["work"]), int(r["fan_ix"])): r for r in b}
for k in sorted(set(ka) | set(kb)):
    ra, rb = ka.get(k), kb.get(k)
    if ra is None or rb is None or ra.tobytes() != rb.tobytes():
        print(k, "emap:", ra, "walk:", rb)
        p = int(off[k[0]]) + k[1]
        print("  fan ids around:", tok[p - n:p + n].tolist())
