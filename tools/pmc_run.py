#!/usr/bin/env python3
"""A handful of single searches of one c2 batch (one lane, records left in HBM): the program
rocprofv3 --pmc runs to count what k_scan_rows does (tools/collect_pmc.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FS_LANES", "1")


def main():
    import torch
    from fandom_search_amd import abi, synth, vocab
    from fandom_search_amd.engine import ScriptIndex
    window = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    conf = synth.CONFIGS["c2"]
    words, emb = synth.vocab_words(), synth.embedding()
    script = synth.script_tokens(conf["script_tokens"])
    swords = [words[int(t)] for t in script]
    chars, coff = vocab.pack_strings(words)
    ix = ScriptIndex(script, swords, emb, synth.lsh_normals(window), cfg=abi.make_config(window_size=window))
    tok, off = synth.corpus_tokens(conf["n_works"], conf["tokens_per_work"], script)
    c = ix.corpus(tok, off, chars, coff)
    rows, st = ix.search(c)
    cap = len(rows) * 2 + 64
    buf = torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda")
    for i in range(reps):
        ix.search_end(ix.search_begin(c, buf.data_ptr(), cap, header=True))
    print("kernel", ix.kernel_name(c), "rows", len(rows))


if __name__ == "__main__":
    main()
