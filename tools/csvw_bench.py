#!/usr/bin/env python3
"""Time of the native batch-file formatter by itself (no GPU): 15 000 records of a 500-work batch against a
20 000-token script, fs_csvw_format on one thread.

  python tools/csvw_bench.py
"""
import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fandom_search_amd import csvw, abi, synth, vocab
words = synth.vocab_words()
n_script = 20000
sw = [words[i % len(words)] for i in range(n_script)]
w = csvw.CsvWriter(sw, [vocab.hash_string(x) for x in sw], ["CHAR%d" % (i % 8) for i in range(n_script)], [i // 500 for i in range(n_script)], words, writers=1)
n = 15000
rng = np.random.default_rng(0)
rows = np.zeros(n, dtype=abi.ROW_DTYPE)
rows['work'] = np.sort(rng.integers(0, 500, n)); rows['fan_ix'] = rng.integers(0, 2000, n); rows['orig_ix'] = rng.integers(0, n_script, n); rows['lev'] = 7
sids = rng.integers(0, len(words), n).astype(np.uint32)
names = ["/tmp/somewhere/fan/w%07d.txt" % i for i in range(500)]
w.format(names, rows, sids)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); b = w.format(names, rows, sids); ts.append(time.perf_counter() - t0)
print("format ms", round(min(ts) * 1e3, 2), "median", round(sorted(ts)[10] * 1e3, 2), "bytes", len(b))
