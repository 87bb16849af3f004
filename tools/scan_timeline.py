#!/usr/bin/env python3
"""Where a k_scan_rows launch spends its time: in-kernel stamps (FS_DIAG=2) per wave range.

  python tools/scan_timeline.py [--workload c2] [--rotate 4] [--extra "FS_X=1 FS_Y=2"]

Runs a few searches (one at a time, ids from HBM when --rotate batches exceed the Infinity
Cache), reads the stamps of the last one through fs_debug_stamps and prints, in
microseconds from the first wave's entry, the percentiles of: entry, filter staged, scan
done (before the range's last flush), rounds done, finished; and the rounds / flushes per
range.  Diagnostic build paths only: the stamps cost a few scalar instructions.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--window", type=int, default=6)
    ap.add_argument("--rotate", type=int, default=4)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--extra", default="")
    ap.add_argument("--dump", default="", help="write the raw stamps (.npy) here")
    ap.add_argument("--load", default="", help="summarise stamps saved by --dump (no GPU needed)")
    ap.add_argument("--kernel", default="", help="with --load: the kernel's name, for the summary")
    ap.add_argument("--kernel-us", type=float, default=0.0, help="with --load: its duration by events")
    a = ap.parse_args()
    os.environ["FS_DIAG"] = str(2 | int(os.environ.get("FS_DIAG", "0")))
    os.environ.setdefault("FS_LANES", "1")
    for kv in a.extra.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
    if a.load:
        d = np.load(a.load).astype(np.int64)
        kernel_name, kernel_us = a.kernel or "(saved stamps)", a.kernel_us
    else:
        import torch
        from fandom_search_amd import _lib, abi, synth, vocab
        from fandom_search_amd.engine import ScriptIndex
        conf = synth.CONFIGS[a.workload]
        words, emb = synth.vocab_words(), synth.embedding()
        script = synth.script_tokens(conf["script_tokens"])
        swords = [words[int(t)] for t in script]
        chars, coff = vocab.pack_strings(words)
        ix = ScriptIndex(script, swords, emb, synth.lsh_normals(a.window), cfg=abi.make_config(window_size=a.window))
        corpora = []
        for r in range(a.rotate):
            t, o = synth.corpus_tokens(conf["n_works"], conf["tokens_per_work"], script,
                                       first_work=r * conf["n_works"])
            corpora.append(ix.corpus(t, o, chars, coff))
        rows, st = ix.search(corpora[0])
        cap = len(rows) * 2 + 64
        buf = torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda")
        ix.set_scan_timing(1)
        ms = []
        for i in range(a.reps * a.rotate):
            n, st = ix.search_end(ix.search_begin(corpora[i % a.rotate], buf.data_ptr(), cap, header=True))
            ms.append(st.scan_ms)
        L = _lib.load()
        n = C.c_uint64(0)
        _lib.check(L.fs_debug_stamps(ix._h, 0, None, 0, C.byref(n)), "fs_debug_stamps")
        out = np.zeros(n.value, dtype=np.uint64)
        _lib.check(L.fs_debug_stamps(ix._h, 0, out.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)),
                   "fs_debug_stamps")
        d = out.reshape(-1, 20).astype(np.int64)
        kernel_name = ix.kernel_name(corpora[0])
        kernel_us = round(float(np.mean(ms[a.rotate:])) * 1e3, 2)
    if a.dump:
        np.save(a.dump, d)
    t0 = d[:, 0].min()
    us = (d[:, :5] - t0) / 100.0
    names = ["entry", "ready", "scan_done", "rounds_done", "finished"]
    res = {"kernel": kernel_name, "kernel_us_events": kernel_us,
           "ranges": int(len(d)), "extra": a.extra}
    pct = [0, 10, 50, 90, 100]
    for k, nm in enumerate(names):
        res[nm] = [round(float(x), 2) for x in np.percentile(us[:, k], pct)]
    res["scan_dur"] = [round(float(x), 2) for x in np.percentile(us[:, 2] - us[:, 1], pct)]
    res["rounds_dur"] = [round(float(x), 2) for x in np.percentile(us[:, 3] - us[:, 2], pct)]
    res["finish_dur"] = [round(float(x), 2) for x in np.percentile(us[:, 4] - us[:, 3], pct)]
    if d[:, 14].max() > 0:
        fz = (d[:, 14:17] - t0) / 100.0
        res["workgroup_together"] = [round(float(x), 2) for x in np.percentile(fz[:, 0], pct)]
        polling = (np.arange(len(d)) % 16) < 4          # the waves that ask for the counts in front
        res["counts_known_to_wave"] = [round(float(x), 2) for x in np.percentile(fz[polling, 1], pct)]
        res["counts_known_to_workgroup"] = [round(float(x), 2) for x in np.percentile(fz[:, 2], pct)]
    if len(d) % 16 == 0:
        # workgroup by workgroup (sixteen wave ranges each): which slot scans last, and what stands
        # between the last scanner's last sub-tile and the workgroup's barrier
        wg = us.reshape(-1, 16, 5)
        last_scan = wg[:, :, 2].max(axis=1)
        res["scan_done_by_slot_mean"] = [round(float(x), 1) for x in wg[:, :, 2].mean(axis=0)]
        res["last_scanner_slot_hist"] = [int(x) for x in np.bincount(wg[:, :, 2].argmax(axis=1), minlength=16)]
        res["workgroup_last_scan"] = [round(float(x), 2) for x in np.percentile(last_scan, pct)]
        res["workgroup_rounds_done"] = [round(float(x), 2) for x in np.percentile(wg[:, :, 3].max(axis=1), pct)]
        res["workgroup_rounds_behind_last_scan"] = [round(float(x), 2) for x in
                                                    np.percentile(wg[:, :, 3].max(axis=1) - last_scan, pct)]
        rr = d[:, 5].reshape(-1, 16)
        res["rounds_of_last_scanner"] = {int(k): int(v) for k, v in zip(*np.unique(
            rr[np.arange(len(rr)), wg[:, :, 2].argmax(axis=1)], return_counts=True))}
    res["rounds_per_range"] = {int(k): int(v) for k, v in zip(*np.unique(d[:, 5], return_counts=True))}
    res["flushes_per_range"] = {int(k): int(v) for k, v in zip(*np.unique(d[:, 6], return_counts=True))}
    res["records_per_range"] = [int(x) for x in np.percentile(d[:, 7], [0, 50, 100])]
    # phase sums of the rounds (the diagnostic build drains its loads at the phase ends, so
    # read the shares, not the lengths), microseconds per ROUND, mean over the ranges
    phases = ["pick", "ids_arrive", "table_arrives", "hits_stored", "records_emitted", "carry"]
    rounds = np.maximum(d[:, 5], 1)
    res["round_phase_us"] = {nm: round(float(np.mean(d[:, 8 + k] / rounds)) / 100.0, 3) for k, nm in enumerate(phases)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
