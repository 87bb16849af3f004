#!/usr/bin/env python3
"""End-to-end time of the reference CLI command on synthetic files (diagnostic).

  python tools/cli_bench.py [--works 5000] [--tokens 2000]

Writes the works as text files and the script in the reference's markup into a
temporary directory, runs `ao3.py search <dir> <script>` there and reports where
the wall time goes.  The GPU part of a 500-work batch is a fraction of a
millisecond; the command is bound by reading, tokenising and encoding the text
and by writing the CSVs on the host."""

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fandom_search_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--works", type=int, default=5000)
    ap.add_argument("--tokens", type=int, default=2000)
    ap.add_argument("--prose", action="store_true",
                    help="write the works as prose: sentences of 4-18 words with a capital first letter and "
                         "closing punctuation, commas, quotes and contractions (the tokenizer's rule path, "
                         "string ids next to vector ids, out-of-vocabulary tokens: the LSH pipeline)")
    a = ap.parse_args()
    words = synth.vocab_words()
    script = synth.script_tokens(20000)
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        fan = os.path.join(tmp, "fan")
        names = synth.write_corpus(fan, a.works, a.tokens, script, words)
        if a.prose:
            import random
            rnd = random.Random(5)
            for path in names:
                with open(path, encoding="utf8") as fh:
                    toks = fh.read().split(" ")
                out, i = [], 0
                while i < len(toks):
                    sent = toks[i:i + rnd.randint(4, 18)]
                    i += len(sent)
                    sent[0] = sent[0].capitalize()
                    if rnd.random() < 0.3:
                        sent[rnd.randrange(len(sent))] += ","
                    if rnd.random() < 0.15:
                        sent.insert(rnd.randrange(len(sent)), "don't")
                    sent[-1] += rnd.choice([".", "?", "!"])
                    if rnd.random() < 0.2:
                        sent[0] = '"' + sent[0]
                        sent[-1] += '"'
                    out += sent
                with open(path, "w", encoding="utf8") as fh:
                    fh.write(" ".join(out))
        spath = os.path.join(tmp, "script.txt")
        with open(spath, "w", encoding="utf8") as fh:
            fh.write(synth.script_markup(script, words))
        t_write = time.time() - t0
        t0 = time.time()
        out = subprocess.run([sys.executable, os.path.join(ROOT, "ao3.py"), "search", fan, spath, "--synthetic-vocab"],
                             cwd=tmp, capture_output=True, text=True, env=dict(os.environ, FANDOM_SEARCH_TIMING="1"))
        dt = time.time() - t0
        csvs = [f for f in os.listdir(tmp) if f.startswith("match-")]
        rows = 0
        for f in csvs:
            if "batch" not in f:
                with open(os.path.join(tmp, f)) as fh:
                    rows = sum(1 for _ in fh) - 1
        print(json.dumps({"works": a.works, "tokens_per_work": a.tokens, "prose": bool(a.prose), "write_inputs_s": round(t_write, 2),
                          "search_command_s": round(dt, 2), "works_per_s": round(a.works / dt, 1),
                          "rows": rows, "csv_files": len(csvs), "rc": out.returncode,
                          "stderr_tail": out.stderr[-900:]}))


if __name__ == "__main__":
    main()
