import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["FS_LANES"] = "1"
import torch
from fandom_search_amd import abi, synth, vocab
from fandom_search_amd.engine import ScriptIndex
conf = synth.CONFIGS["c2"]
words, emb = synth.vocab_words(), synth.embedding()
script = synth.script_tokens(conf["script_tokens"])
swords = [words[int(t)] for t in script]
chars, coff = vocab.pack_strings(words)
ix = ScriptIndex(script, swords, emb, synth.lsh_normals(6), cfg=abi.make_config())
cs = []
for r in range(4):
    t, o = synth.corpus_tokens_parallel(conf["n_works"], conf["tokens_per_work"], script, first_work=r * conf["n_works"])
    cs.append(ix.corpus(t, o, chars, coff))
rows, st = ix.search(cs[0])
cap = len(rows) + 4096
bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda") for _ in range(5)]
for period in (4, 1):
    ix.set_scan_timing(period)
    for total in (80, 400, 2000):
        one, tk = [], []
        torch.cuda.synchronize(); time.sleep(0.05)
        for i in range(total):
            tk.append(ix.search_begin(cs[i % 4], bufs[i % 5].data_ptr(), cap, header=True))
            if len(tk) >= 3:
                one.append(ix.search_end(tk.pop(0))[1].scan_ms)
        while tk:
            one.append(ix.search_end(tk.pop(0))[1].scan_ms)
        v = np.array([x for x in one if x > 0]) * 1e3
        q = len(v) // 4
        print(json.dumps(dict(period=period, total=total, first_quarter=round(float(v[:q].mean()), 2), last_quarter=round(float(v[-q:].mean()), 2), mean=round(float(v.mean()), 2), mn=round(float(v.min()), 2))), flush=True)
