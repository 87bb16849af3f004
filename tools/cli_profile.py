#!/usr/bin/env python3
"""cProfile of `ao3.py search` on 3000 synthetic files (diagnostic): where the host time of the
reference's command goes (start-up and library load, reading + tokenising, encoding, CSVs)."""
import cProfile, pstats, os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fandom_search_amd import synth, cli
words = synth.vocab_words()
script = synth.script_tokens(20000)
tmp = tempfile.mkdtemp()
fan = os.path.join(tmp, "fan")
synth.write_corpus(fan, 3000, 2000, script, words)
spath = os.path.join(tmp, "script.txt")
open(spath, "w", encoding="utf8").write(synth.script_markup(script, words))
os.chdir(tmp)
t0 = time.time()
cProfile.run("cli.main(['search', fan, spath, '--synthetic-vocab'])", "/tmp/cli.prof")
print("total", time.time() - t0)
pstats.Stats("/tmp/cli.prof").sort_stats("cumtime").print_stats(28)
