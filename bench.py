#!/usr/bin/env python3
"""fanworks/sec of the 6-gram search hot path on MI355X.

A step = one pass of the search over one batch of synthetic fan works that is already
resident in HBM (fs_search_corpus_begin/_end of include/fandom_search.h: scan, exact
verification, Levenshtein records, per-word dedupe, records left in HBM).

  N = 1   BASELINE.json configs[1] ("c2": 10k works x 2k tokens vs a 2k-line script,
          6-gram).  The timed steps rotate over four DISTINCT c2 batches (320 MB of
          ids, more than the 256 MiB Infinity Cache), so every step reads its ids
          from HBM; `--rotate 1` re-scans one resident batch (bound: infinity-cache).
  N > 1   BASELINE.json configs[2] ("c3": 100k works x 5k tokens) split N ways (strong
          scaling: rank r holds works [r W/N, (r+1) W/N), one GPU each), the match
          records of all ranks delivered over RCCL inside the step to rank (step mod N):
          one all_to_all per N steps, every pair of GPUs having its own xGMI link
          (--gather-root step: a gather per step; --gather-root 0: always to rank 0)
          (fandom_search_amd.dist.RowGather: 8-byte wire records, count and records in
          one collective, gather of step i beside the search of step i+1).  The N = 1
          line carries the same corpus as companions.c3_full (eight 12.5k-work shards
          on one GPU), so 1 -> 8 is one corpus.  `--scaling weak`: one c2 batch per rank.
          `python bench.py --gpus N` typed as is starts the N ranks itself
          (torch.distributed.run as a child process); under a launcher (WORLD_SIZE set)
          it is one of the ranks.

The timed region is `--steps` steps between barrier + synchronize on both sides; when
steps <= 64 it is repeated (9 regions, each primed by the warm-up) and the median region
is reported, with every sample in `samples_ms` (a 0.6 ms region is a fragile basis).

Prints ONE JSON line (rank 0): the driver's contract plus
  roofline      the dominant kernel (k_scan_rows: tokens -> records in one launch):
                algorithmic bytes per launch (4 B per fan token + 32 B per record,
                SURVEY 8(d)) / its average duration over the timed steps, from HIP
                events attached to its dispatch; `step` = the same bytes / ms_per_step;
                `alone_ms` = its duration with nothing else on the GPU (measured after
                the timed region), since several searches in flight share the machine
  companions    (N = 1) the same search with what `value` leaves out: records copied
                to the host, ids streamed from pinned host memory, a mixed-case fan
                side (per-match Levenshtein on the GPU), and the LSH pipeline on a
                table with near-synonyms (where the exact-n-gram proof fails)
  cpu_baseline  the plain-C oracle (the reference's LSH algorithm) on a bounded sample
                of the same workload on the host cores, and as `reference_shaped` the
                literal Python restatement on configs[0] in a 4-process pool, the way
                search.py:381-385 runs the original
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
MALL_BYTES = 256 << 20    # Infinity Cache


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="",
                    help="c2 (default at N=1), c3 (default at N>1, split over the ranks), c3shard, c1")
    ap.add_argument("--scaling", default="",
                    help="N > 1: strong (default: configs[2] = c3 split N ways) or weak (one c2 batch per rank)")
    ap.add_argument("--gather-root", default="rotate",
                    help="N > 1: rotate (step i's records to rank i mod N, one all_to_all per N steps; default), "
                         "step (the same with one gather per step) or 0 (a gather per step, always to rank 0)")
    ap.add_argument("--reps", type=int, default=0,
                    help="timed regions of --steps steps (default: 9 when steps <= 64, else 3); "
                         "the median is reported")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch: rendezvous port (default: a free one)")
    ap.add_argument("--works", type=int, default=0, help="override works (per rank)")
    ap.add_argument("--window", type=int, default=6)
    ap.add_argument("--rotate", type=int, default=0,
                    help="distinct batches the timed steps rotate over (default: enough "
                         "to exceed the Infinity Cache, at most 4)")
    ap.add_argument("--lanes", type=int, default=4,
                    help="streams the library spreads searches over (FS_LANES): searches in "
                         "flight overlap on the GPU; 1 = one after the other")
    ap.add_argument("--inflight", type=int, default=0, help="searches kept in flight (default: lanes)")
    ap.add_argument("--scan-timing", type=int, default=0,
                    help="HIP events around the dominant kernel of every Nth search (default: every "
                         "search when steps <= 64, every 4th otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-shaped", action="store_true",
                    help="skip the Python reference-shaped leg of cpu_baseline")
    ap.add_argument("--no-companions", action="store_true")
    ap.add_argument("--backend", default="nccl",
                    help="nccl (RCCL, one GPU per rank) or gloo (rehearsal: every rank "
                         "computes on GPU 0, records gathered through host memory)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--wire", type=int, default=0, help="N > 1: force 16-byte wire records")
    return ap.parse_args(argv)


def launcher_command(args, argv, env, port=None):
    """`python bench.py --gpus N` typed without a launcher: the command that starts the N
    ranks (None when this process is a rank already, or N == 1).  The parent never touches
    the GPU; the ranks run as a child process whose exit code it returns."""
    if args.gpus <= 1:
        return None
    # under a launcher: RANK / LOCAL_RANK are set next to WORLD_SIZE (an image that merely
    # exports WORLD_SIZE=1 has not launched anything)
    if "WORLD_SIZE" in env and ("RANK" in env or "LOCAL_RANK" in env or env["WORLD_SIZE"] != "1"):
        return None
    if port is None:
        port = args.master_port
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus)]
    if port:
        head += ["--master-addr", "127.0.0.1", "--master-port", str(port)]
    else:
        # no port named: the launcher's own rendezvous picks a free one (nothing to collide on
        # between concurrent runs of one box)
        head += ["--standalone", "--local-addr", "127.0.0.1"]
    return head + [os.path.abspath(__file__)] + list(argv)


def host_cores():
    """CPU cores this process may actually use: the affinity mask capped by the
    cgroup CPU quota (a GPU box exposes all host CPUs but grants a share)."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("FS_BENCH_CORES")
    return int(env) if env else cores


def cpu_baseline(cfg, script, swords, words, emb, normals, tokens_per_work,
                 chars, coff, budget_s):
    """Plain-C oracle (reference algorithm: LSH keys, bucket candidates, cosine,
    top-10, Levenshtein, per-word dedupe) on the first works of the workload."""
    from fandom_search_amd import synth, vocab
    from oracle import c_oracle
    cores = host_cores()
    sch, so = vocab.pack_strings(swords)
    oi = c_oracle.OracleIndex(cfg, script, sch, so, emb, normals, threads=cores)
    pilot = 4 * cores
    tok, off = synth.corpus_tokens(pilot, tokens_per_work, script)
    oi.search(tok, off, chars, coff)             # also fills the per-token tables
    t0 = time.perf_counter()
    oi.search(tok, off, chars, coff)
    per_work = (time.perf_counter() - t0) / pilot
    n = int(max(pilot, min(4000, budget_s / max(per_work, 1e-9))))
    tok, off = synth.corpus_tokens(n, tokens_per_work, script)
    t0 = time.perf_counter()
    rows, st = oi.search(tok, off, chars, coff)
    dt = time.perf_counter() - t0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n / dt, "unit": "fanworks/s", "cores": cores, "cpu_model": model, "kind": "port",
            "sample": "first %d works (%d tokens each) of the workload, %.1f s, "
                      "oracle/fs_oracle.c with OpenMP over works"
                      % (n, tokens_per_work, dt)}


_REF = {}


def _ref_search(w):
    """One fan work through the literal restatement (pool worker)."""
    return len(_REF["idx"].search("w%07d.txt" % w, _REF["toks"](_REF["works"][w])))


def cpu_reference_shaped(window):
    """the literal Python/numpy restatement of search.py driven the
    way search.py:381-385 drives the original: Pool(4).map(..., chunksize=31) over the
    works of configs[0] (c1: 50 works x 1000 tokens, 5000-token script).  Runs before
    anything touches the GPU (the pool forks)."""
    import multiprocessing as mp
    from fandom_search_amd import abi, synth, vocab
    from oracle import nearpy_restated as nr
    from oracle import search_restated as sr
    conf = synth.CONFIGS["c1"]
    words, emb = synth.vocab_words(), synth.embedding()
    voc = vocab.Vocab(words, emb)
    normals = synth.lsh_normals(window)
    script = synth.script_tokens(conf["script_tokens"])
    scene, char = synth.script_columns(len(script))
    tok, off = synth.corpus_tokens(conf["n_works"], conf["tokens_per_work"], script)

    def toks(ids):
        return [sr.Tok(words[i], voc.orth(i), words[i], voc.orth(i), emb[i]) for i in ids]

    rows = [[words[t], voc.orth(t), int(scene[i]), char[i]] for i, t in enumerate(script)]
    t0 = time.perf_counter()
    _REF["idx"] = sr.AnnIndexSearch(rows, toks(script), window, 15, 14, 0.1, normals,
                                    arith=nr.LiteralArith(), unique_filter=abi.default_unique_filter())
    t_index = time.perf_counter() - t0
    _REF["toks"] = toks
    _REF["works"] = [tok[int(off[w]):int(off[w + 1])] for w in range(conf["n_works"])]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(processes=4) as pool:
        n_rows = sum(pool.map(_ref_search, range(conf["n_works"]), chunksize=31))
    dt = time.perf_counter() - t0
    _REF.clear()
    return {"value": conf["n_works"] / dt, "unit": "fanworks/s", "processes": 4,
            "chunksize": 31, "workload": "c1", "rows": n_rows, "index_s": round(t_index, 2),
            "sample": "all %d works of c1 (%d tokens each), %.1f s, literal Python/numpy "
                      "restatement (oracle/search_restated.py) in Pool(4), chunksize 31 "
                      "= %d busy workers" % (conf["n_works"], conf["tokens_per_work"], dt,
                                             -(-conf["n_works"] // 31))}


def search_companion(ix, corpus, n_works, n_tok, n_fl, reps=24, alone_reps=6):
    """One workload through one index, measured the way the headline is (as many searches in
    flight as the timed region keeps, records left in HBM) and by itself, with the per-kernel
    times of one search (fs_search_profile).  `roofline` is the WHOLE search: algorithmic bytes
    (4 B per fan token, + 32 B per record where one kernel takes tokens in and puts records
    out) / the device time of one search alone / HBM peak, and `kernel` is the kernel with the
    largest share of that search -- for a search of several launches no single kernel's own
    fraction stands for it (the prefilter scan's is kept as `prefilter_frac`)."""
    import torch
    from fandom_search_amd import abi
    rows, st = ix.search(corpus)
    best = None
    for _ in range(alone_reps):
        rows, st = ix.search(corpus, reuse=True)
        if st.total_ms > 0:           # (0: the search was repeated to grow a workspace)
            best = st.total_ms if best is None else min(best, st.total_ms)
    cap = len(rows) + 64
    bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda") for _ in range(n_fl + 1)]
    torch.cuda.synchronize()      # (torch's fill is complete before the library's own streams write the buffer)
    for i in range(2 * n_fl):                                    # (every lane's workspaces grown)
        ix.search_end(ix.search_begin(corpus, bufs[0].data_ptr(), cap, header=True))
    ix.set_scan_timing(1 << 20)
    samples = []
    for _ in range(3):
        tickets = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            tickets.append(ix.search_begin(corpus, bufs[i % len(bufs)].data_ptr(), cap, header=True))
            if len(tickets) >= n_fl:
                ix.search_end(tickets.pop(0))
        while tickets:
            ix.search_end(tickets.pop(0))
        torch.cuda.synchronize()
        samples.append((time.perf_counter() - t0) / reps)
    dt = float(np.median(samples))
    ix.set_scan_timing(1)
    prof, prof_sums = None, []
    for _ in range(4):                                            # (the last one: warm)
        prof = ix.profile(corpus, bufs[0].data_ptr() + 32, cap)
        prof_sums.append(sum(ms for _, ms in prof))
    if st.path == abi.FS_MODE_EXACT:
        # (the synchronous call above delivers host rows, which the exact pipeline's last kernel
        # stores straight into pinned host memory: PCIe inside its time.  A search alone with the
        # records left in HBM, as everywhere else on this line: the profiled one)
        best = min(prof_sums[1:])
    del bufs
    kernel = ix.kernel_name(corpus)
    fused_rows = kernel.startswith("k_scan_rows")
    algo = 4.0 * n_tok + (32.0 * len(rows) if fused_rows else 0.0)
    tot = sum(ms for _, ms in prof) or 1e-9
    dom, dom_ms = max(prof, key=lambda kv: kv[1])
    first_ms = prof[0][1]
    out = {"value": n_works / dt, "unit": "fanworks/s", "ms_per_step": dt * 1e3,
           "ms_per_search_alone": best, "value_alone": n_works / (best * 1e-3) if best else None,
           "kernel": kernel, "rows_per_step": int(len(rows)), "candidates": int(st.candidates),
           "lsh_pending": int(st.lsh_pending), "matches": int(st.matches),
           "kernels_us": [[k, round(ms * 1e3, 1)] for k, ms in prof],
           "roofline": {"bound": "hbm", "kernel": dom, "kernel_share_of_search": round(dom_ms / tot, 3),
                        "bytes_model": "4 B per fan token" + (" + 32 B per record" if fused_rows else "") +
                                       ", over the whole search (all its kernels), one search alone",
                        "algorithmic_bytes_per_search": algo, "search_ms_alone": best,
                        "achieved": algo / (best * 1e-3) / 1e9 if best else None, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": algo / (best * 1e-3) / 1e9 / HBM_PEAK_GBS if best else None,
                        "step": algo / dt / 1e9 / HBM_PEAK_GBS,
                        "prefilter_frac": (4.0 * n_tok / (first_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                           if len(prof) > 1 and first_ms > 0 else None),
                        "note": "kernels_us: time between HIP events behind consecutive kernels of one search "
                                "(each includes ~5-10 us of event and launch gap); frac: whole search alone; "
                                "step: the same bytes / ms_per_step (searches in flight)"}}
    return out, rows, st


def companions(ix, corpora, toks, offs, chars, coff, words, script, swords, emb, window, n_works):
    """What `value` leaves out, each on a bounded run of the same c2 batches (N = 1)."""
    import torch
    from fandom_search_amd import abi, synth, vocab
    from fandom_search_amd.engine import PinnedBuffer, ScriptIndex, search_stream
    out = {}
    ix.set_scan_timing(1 << 20)       # (no timing events on the companions' launches)
    # (1) records copied to the host after every search (FS_ROWS_HOST, synchronous)
    reps = 10
    ix.search(corpora[0], reuse=True)
    t0 = time.perf_counter()
    for i in range(reps):
        rows, st = ix.search(corpora[i % len(corpora)], reuse=True)
    dt = time.perf_counter() - t0
    out["rows_to_host"] = {"value": n_works * reps / dt, "unit": "fanworks/s",
                           "ms_per_step": dt / reps * 1e3, "rows_per_step": int(len(rows)),
                           "note": "synchronous fs_search_corpus with host rows: 8-byte records over PCIe "
                                   "into pinned memory, fs_row made on the host cores, into a row "
                                   "buffer the caller reuses"}
    # (2) ids streamed from pinned host memory, upload of batch i+1 beside the search of
    # batch i (configs[4] mechanics on c2 batches; PCIe-inclusive)
    pins = []
    for t in toks:
        pb = PinnedBuffer(len(t), np.uint32)
        pb.array[:] = t
        pins.append(pb)
    poffs = []
    for o in offs:
        po = PinnedBuffer(len(o), np.uint64)
        po.array[:] = o
        poffs.append(po)

    def batches(k):
        for i in range(k):
            yield pins[i % len(pins)].array, poffs[i % len(poffs)].array

    for _ in search_stream(ix, batches(2), chars, coff):
        pass
    k = 12
    t0 = time.perf_counter()
    n_rows = 0
    for rows, st in search_stream(ix, batches(k), chars, coff):
        n_rows += len(rows)
    dt = time.perf_counter() - t0
    out["streamed"] = {"value": n_works * k / dt, "unit": "fanworks/s", "ms_per_step": dt / k * 1e3,
                       "ids_GBps": sum(len(toks[i % len(toks)]) for i in range(k)) * 4 / dt / 1e9,
                       "note": "%d c2 batches from pinned host memory through two device "
                               "corpora (fs_corpus_update_begin/_end), records copied back: "
                               "H2D and D2H inside the time" % k}
    for pb in pins + poffs:
        pb.close()
    # (3) mixed-case fan side: string ids differ from vector ids (the reference's fan side is
    # case-sensitive, search.py:166), so every match takes its Levenshtein distance on the GPU
    strings = list(words) + [w.capitalize() for w in words]
    schars, scoff = vocab.pack_strings(strings)
    rng = np.random.default_rng(11)
    tok_str = toks[0].copy()
    cap_sel = rng.random(len(tok_str)) < 0.08
    tok_str[cap_sel] += np.uint32(len(words))
    cs = ix.corpus(toks[0], offs[0], schars, scoff, tok_str=tok_str)
    rows, st = ix.search(cs)
    cap = len(rows) + 64
    n_fl = max(2, int(os.environ.get("FS_LANES", "1")))          # searches in flight, as in the timed region
    bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda") for _ in range(n_fl + 1)]
    torch.cuda.synchronize()      # (torch's fill is complete before the library's own streams write the buffer)
    reps, tickets = 48, []
    for i in range(8):                                           # (primed pipeline)
        ix.search_end(ix.search_begin(cs, bufs[0].data_ptr(), cap, header=True))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        tickets.append(ix.search_begin(cs, bufs[i % len(bufs)].data_ptr(), cap, header=True))
        if len(tickets) >= n_fl:
            ix.search_end(tickets.pop(0))
    while tickets:
        ix.search_end(tickets.pop(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["tok_str"] = {"value": n_works * reps / dt, "unit": "fanworks/s", "ms_per_step": dt / reps * 1e3,
                      "rows_per_step": int(len(rows)), "kernel": ix.kernel_name(cs),
                      "note": "8 % of the fan tokens capitalised: string ids next to vector ids; a hit "
                              "whose tokens are not all written as their vector row's word takes its "
                              "Levenshtein distances inside the same kernel (`kernel`), as many searches in flight as the timed region keeps"}
    cs.close()
    # (4) the LSH pipeline on a table with near-synonyms (c_max ~ 1: the exact-n-gram proof
    # fails, every real embedding table is of this kind), under both settings of NearPy's
    # UniqueFilter (default: off, what NearPy 1.0.0 does for the reference's call)
    emb_c, perm = synth.clustered_table()
    n_l = n_works
    tk = synth.synonym_swaps(toks[0][:n_l * (len(toks[0]) // n_works)], perm)
    of = offs[0][:n_l + 1]
    for name, uf in (("lsh_clustered_table", None), ("lsh_clustered_table_unique_filter_on", True)):
        t0 = time.perf_counter()
        ixl = ScriptIndex(script, swords, emb_c, synth.lsh_normals(window),
                          cfg=abi.make_config(window_size=window, unique_filter=uf))
        t_index = time.perf_counter() - t0
        cl = ixl.corpus(tk, of, chars, coff)
        rec, rows, st = search_companion(ixl, cl, n_l, len(tk), n_fl)
        rec.update({"windows_per_s": st.windows_processed / (rec["ms_per_search_alone"] * 1e-3), "works": n_l,
                    "inexact_rows": int((np.abs(rows["dist"]) > 1e-9).sum()), "c_max": ixl.info["c_max"],
                    "path": "lsh", "index_s": round(t_index, 2),
                    "unique_filter": bool(abi.default_unique_filter() if uf is None else uf),
                    "note": "synth.clustered_table (1024 groups of 8 near-synonyms), 10 % of the fan tokens "
                            "swapped for a synonym; the LSH pipeline behind the component-id prefilters (`kernel`: "
                            "the first kernel of the search; k_lsh_scan when they do not apply); value: as many "
                            "searches in flight as the timed region keeps, like the headline; *_alone: device "
                            "time of one search by itself"})
        out[name] = rec
        cl.close()
        ixl.close()
    # (5) BASELINE.json configs[3]: the n = 4 / 8 / 10 sweep on the 10k-work corpus (n = 4: exact
    # pipeline; n = 8, 10: the proof fails by one slot, LSH pipeline behind the integer prefilters)
    if window == 6:
        for n in (4, 8, 10):
            t0 = time.perf_counter()
            ixn = ScriptIndex(script, swords, emb, synth.lsh_normals(n), cfg=abi.make_config(window_size=n))
            t_index = time.perf_counter() - t0
            cn = ixn.corpus(toks[0], offs[0], chars, coff)
            rec, rows, st = search_companion(ixn, cn, n_works, len(toks[0]), n_fl)
            rec.update({"window": n, "works": n_works, "index_s": round(t_index, 2),
                        "path": "exact-ngram-scan" if st.path == abi.FS_MODE_EXACT else "lsh",
                        "note": "configs[3]: the %d-gram search of the same 10k-work batch, four searches in "
                                "flight (value) and one alone (roofline)" % n})
            out["c4_n%d" % n] = rec
            cn.close()
            ixn.close()
        # (6) the headline's search with NearPy's UniqueFilter on (the exact pipeline's records do not
        # depend on it -- tests/test_golden.py -- and neither does its speed)
        ixu = ScriptIndex(script, swords, emb, synth.lsh_normals(window),
                          cfg=abi.make_config(window_size=window, unique_filter=True))
        cu = ixu.corpus(toks[0], offs[0], chars, coff)
        rec, rows, st = search_companion(ixu, cu, n_works, len(toks[0]), n_fl)
        rec.update({"unique_filter": True, "note": "the headline's batch with unique_filter = 1 (NearPy 0.2.x); "
                                                   "the line itself runs the default, 0"})
        out["headline_unique_filter_on"] = rec
        cu.close()
        ixu.close()
        # (7) a table shaped like a real one (synth.realistic_table: 20k unnormalised rows, similarity at
        # three scales, duplicate and zero rows), fan text with near-synonyms, capitalised words and --
        # first entry -- 8 % out-of-vocabulary names.  What every user's own table looks like: the
        # integer prefilters of the benchmark tables do not apply (norms spread over a factor of ten:
        # "at most one slot may differ" is false), the share rule does (k_share_scan)
        import importlib.util
        spec = importlib.util.spec_from_file_location("realistic_bench", os.path.join(ROOT, "tools", "realistic_bench.py"))
        rb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rb)
        for name, oov in (("lsh_realistic_table", 0.08), ("lsh_realistic_table_no_oov", 0.0)):
            rec = rb.run(works=1000, oov=oov,
                         companion=lambda ixr, cr, w, nt: search_companion(ixr, cr, w, nt, n_fl)[0])
            rec["note"] = ("1000 works x 2000 tokens on synth.realistic_table; value: as many searches in flight "
                           "as the timed region keeps, like the headline; *_alone: device time of one search by "
                           "itself; kernel: the largest share of that search -- k_share_scan = the share rule "
                           "(k_lsh_scan would be the key scan over every window, 5x slower)")
            out[name] = rec
    return out


def command_companion(works=20000, timeout=180):
    """The reference's command itself (`ao3.py search <dir> <script>`, a process of its own) on
    `works` synthetic files of 2000 tokens: wall time from start to the dated CSV, files/s, and
    where it went (tools/cli_bench.py).  Host work -- reading, tokenising, the batch files -- and
    start-up: the GPU's share of a 500-work batch is 0.3 ms.  Never fails the bench line."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_bench.py"), "--works", str(works)],
                           capture_output=True, text=True, timeout=timeout)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": d["works_per_s"], "unit": "fanworks/s", "works": d["works"],
                "tokens_per_work": d["tokens_per_work"], "search_command_s": d["search_command_s"],
                "rows": d["rows"], "csv_files": d["csv_files"], "rc": d["rc"],
                "where": d["stderr_tail"][-700:],
                "note": "python ao3.py search (one process: native text encoder, one GPU, native batch writer) on "
                        "synthetic files written for the run; includes interpreter and HIP start-up"}
    except Exception as e:          # (diagnostic companion: reported, not raised)
        return {"value": None, "error": repr(e)[:300]}


def c3_companions(ix, script, chars, coff, inflight, shards=8, passes=5):
    """BASELINE.json configs[2] (100k works x 5k tokens) on ONE GPU, as the eight 12.5k-work
    shards an 8-GPU run deals out (the script is configs[1]'s, so the index is the same):
      c3shard  a step = one shard (250 MB of ids), rotating over the eight distinct ones
      c3_full  a step = the whole corpus, eight searches (the N = 1 point of the 1 -> 8 curve)"""
    import torch
    from fandom_search_amd import _lib, abi, synth
    conf = synth.CONFIGS["c3"]
    per, tpw = conf["n_works"] // shards, conf["tokens_per_work"]
    t0 = time.perf_counter()
    corpora, rows = [], []
    for k in range(shards):
        t, o = synth.corpus_tokens_parallel(per, tpw, script, first_work=k * per)
        corpora.append(ix.corpus(t, o, chars, coff))
    t_gen = time.perf_counter() - t0
    n_tok = corpora[0].n_tok
    cap = n_tok // 64
    probe = torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # (torch's fill is complete before the library's own streams write the buffer)
    for c in corpora:
        while True:
            try:
                rows.append(ix.search_device(c, probe.data_ptr() + 32, cap)[0])
                break
            except _lib.FsError as e:
                if e.code != abi.FS_E_CAPACITY:
                    raise
                cap = int(e.required * 1.05) + 64
                probe = torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()      # (torch's fill is complete before the library's own streams write the buffer)
    del probe
    bufs = [torch.zeros(32 + cap * 32, dtype=torch.uint8, device="cuda") for _ in range(inflight + 1)]
    torch.cuda.synchronize()      # (torch's fill is complete before the library's own streams write the buffer)

    def run(n_steps):
        tickets = []
        for i in range(n_steps):
            tickets.append(ix.search_begin(corpora[i % shards], bufs[i % len(bufs)].data_ptr(), cap, header=True))
            if len(tickets) >= inflight:
                ix.search_end(tickets.pop(0))
        while tickets:
            ix.search_end(tickets.pop(0))

    ix.set_scan_timing(1 << 20)                   # (no events on these launches: an event record costs stream time)
    run(shards)
    samples = []
    per_region = 6                                # passes per timed region (one pass = the whole corpus)
    for _ in range(passes):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(shards * per_region)
        torch.cuda.synchronize()
        samples.append((time.perf_counter() - t0) / per_region)
    dt = float(np.median(samples))
    ix.set_scan_timing(1)
    alone = [ix.search_end(ix.search_begin(corpora[i % shards], bufs[0].data_ptr(), cap, header=True))[1].scan_ms
             for i in range(10)]
    alone_ms = float(np.mean(alone[2:]))
    shard_bytes = 4.0 * n_tok + 32.0 * float(np.mean(rows))
    total_bytes = 4.0 * n_tok * shards + 32.0 * float(np.sum(rows))
    kernel = ix.kernel_name(corpora[0])
    out = {
        "c3shard": {"value": per * shards / dt, "unit": "fanworks/s", "ms_per_step": dt / shards * 1e3,
                    "tokens_per_s": per * tpw * shards / dt, "works": per, "tokens_per_work": tpw,
                    "rows_per_step": int(round(float(np.mean(rows)))),
                    "roofline": {"bound": "hbm", "kernel": kernel, "launch_ms_alone": alone_ms,
                                 "frac_alone": shard_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "step": shard_bytes / (dt / shards) / 1e9 / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": shard_bytes, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s"},
                    "note": "one GPU's share of configs[2] per step (12.5k works x 5k tokens, 250 MB of ids), "
                            "eight distinct shards in rotation (2 GB), %d searches in flight, records left in HBM"
                            % inflight},
        "c3_full": {"value": conf["n_works"] / dt, "unit": "fanworks/s", "ms_per_step": dt * 1e3,
                    "tokens_per_s": conf["n_works"] * tpw / dt, "works": conf["n_works"],
                    "tokens_per_work": tpw, "launches": shards, "rows": int(np.sum(rows)),
                    "samples_ms": [round(x * 1e3, 4) for x in samples],
                    "roofline_step": total_bytes / dt / 1e9 / HBM_PEAK_GBS,
                    "corpus_s": round(t_gen, 1),
                    "note": "configs[2] whole on one GPU (timed: six passes per region, five regions, the median "
                            "pass): a step = the eight shards of an 8-GPU run searched one behind the other (the N = 1 point of the strong-scaling curve `--gpus N` "
                            "measures; same works, same shards)"},
    }
    for c in corpora:
        c.close()
    return out


def main():
    args = parse()
    cmd = launcher_command(args, sys.argv[1:], os.environ)
    if cmd is not None:
        import subprocess
        if args.backend == "nccl":
            # (device_count() does not initialise the GPU on this image: the parent stays clean)
            import torch
            from fandom_search_amd.dist import check_device_count
            check_device_count(0, args.gpus, torch.cuda.device_count())
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    os.environ["FS_LANES"] = str(max(1, args.lanes))
    inflight = args.inflight or max(1, args.lanes)
    ref_shaped = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline
            and not args.no_reference_shaped):
        ref_shaped = cpu_reference_shaped(args.window)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d under a launcher of %d ranks" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    from fandom_search_amd import _lib, abi, synth, vocab
    from fandom_search_amd.dist import HDR, RowGather
    from fandom_search_amd.engine import ScriptIndex

    rehearsal = args.backend == "gloo"
    dev_index = 0 if rehearsal else local_rank
    if world > 1 and not rehearsal:
        # one process per GPU over RCCL: refuse up front when two ranks would share a device
        # (fandom_search_amd.dist.init_nccl_checked: device count, then identities through the
        # rendezvous store, then the first collective)
        from fandom_search_amd.dist import init_nccl_checked
        init_nccl_checked(local_rank, world)
    else:
        torch.cuda.set_device(dev_index)
        if world > 1:
            dist.init_process_group("gloo")

    # ---- workload -------------------------------------------------------------------
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    if scaling not in ("strong", "weak"):
        raise SystemExit("--scaling strong|weak")
    strong = world > 1 and scaling == "strong"
    wl = args.workload or ("c3" if strong else "c2")
    conf = dict(synth.CONFIGS[wl])
    tpw = conf["tokens_per_work"]
    if strong:
        total_works = args.works * world if args.works else conf["n_works"]
        lo, hi = rank * total_works // world, (rank + 1) * total_works // world
        n_works, first_work = hi - lo, lo
    else:
        n_works = args.works or conf["n_works"]
        total_works = n_works * world
        first_work = rank * n_works
    shard_bytes = 4 * n_works * tpw
    # distinct batches in rotation: as many as it takes to exceed the Infinity Cache (at most 4)
    rotate = args.rotate or min(4, -(-(MALL_BYTES + 1) // shard_bytes))
    rotate = max(1, rotate)
    words = synth.vocab_words()
    emb = synth.embedding()
    normals = synth.lsh_normals(args.window)
    script = synth.script_tokens(conf["script_tokens"])
    swords = [words[int(t)] for t in script]
    chars, coff = vocab.pack_strings(words)
    toks, offs = [], []
    gen_procs = max(2, host_cores() // max(1, world))
    for r in range(rotate):          # distinct works per batch (and per rank)
        t, o = synth.corpus_tokens_parallel(n_works, tpw, script, first_work=first_work + r * total_works,
                                            procs=gen_procs)
        toks.append(t)
        offs.append(o)

    cfg = abi.make_config(window_size=args.window, device=dev_index)
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    corpora = [ix.corpus(t, o, chars, coff) for t, o in zip(toks, offs)]
    n_tok = corpora[0].n_tok

    # wire records of the gather: 8 bytes (exact pipeline, scripts below 2^18 tokens), else
    # 16; fs_row when there is one rank
    packed = False
    if world > 1 and ix.info["path"] == abi.FS_MODE_EXACT:
        packed = 8 if len(script) < abi.PACKED8_MAX_SCRIPT and args.wire != 16 else 16
    rec_bytes = packed if packed else 32
    cap = 4096
    probe = torch.zeros(HDR + cap * rec_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # (torch's fill is complete before the library's own streams write the buffer)
    rows_per_corpus = []
    for c in corpora:
        while True:
            try:
                n, st0 = ix.search_device(c, probe.data_ptr() + HDR, cap, packed=packed)
                rows_per_corpus.append(n)
                break
            except _lib.FsError as e:
                if e.code != abi.FS_E_CAPACITY:
                    raise
                cap = int(e.required * 1.05) + 64
                probe = torch.zeros(HDR + cap * rec_bytes, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()      # (torch's fill is complete before the library's own streams write the buffer)
    del probe
    NB = inflight + 1
    # the receiver of a step's records goes round (step i to rank i mod N): every pair of GPUs
    # has its own xGMI link, and the seven links that end at one GPU carry less than eight
    # searches produce (--gather-root 0: always rank 0)
    any_root = world > 1 and args.gather_root in ("rotate", "step")
    # ... and one all_to_all per N steps carries them (every link of every rank at once) instead
    # of one gather per step (one link per rank and step); --gather-root step: a gather per step
    exchange = world > 1 and args.gather_root == "rotate"
    # (exchange: a buffer is free again when its group's all_to_all has been queued and waited
    # for, i.e. world + inflight - 1 buffers are in use at once, in whole groups)
    gather = RowGather(ix, cap, rec_bytes,
                       n_buffers=max(NB, 3 * world, -(-(world + inflight - 1) // world) * world) if exchange else NB,
                       rehearsal=rehearsal, any_root=any_root, exchange=exchange, inflight=inflight)
    NB = gather.n_buffers
    cap = gather.cap
    if packed == 8:
        gather.set_offsets(offs[0])

    # ---- the step: search of batch i queued while earlier ones run, gathers behind ----
    tickets = {}
    scan_ms = []
    last = {"st": None, "rows": 0, "buf": 0, "timing": False}

    trace = [] if os.environ.get("BENCH_TRACE_HOST") else None

    def complete(i):
        t, b = tickets.pop(i)
        n, st = ix.search_end(t)
        if trace is not None:
            trace.append(("end", i, time.perf_counter(), st.scan_ms))
        if st.scan_ms > 0 and last["timing"]:
            scan_ms.append(st.scan_ms)
        last["st"], last["rows"] = st, n
        gather.start(b, (b % world if exchange else i % world) if any_root else 0)
        last["buf"] = b

    def step(i):
        b = i % NB
        gather.wait(b)
        tickets[i] = (ix.search_begin(corpora[i % rotate], gather.bufs[b].data_ptr(), cap,
                                      packed=packed, header=True), b)
        if trace is not None:
            trace.append(("begin", i, time.perf_counter(), 0.0))
        if i - (inflight - 1) in tickets:
            complete(i - (inflight - 1))

    def drain():
        for i in sorted(tickets):
            complete(i)
        gather.flush()
        for b in range(NB):
            gather.wait(b)

    # the dominant kernel carries timing events on every search of a short run (the driver
    # passes --steps 20), on every 4th of a long one (an event record costs stream time)
    ix.set_scan_timing(args.scan_timing or (1 if args.steps <= 64 else 4))
    reps = args.reps or (9 if args.steps <= 64 else 3)
    samples = []
    it = 0
    for rep in range(reps):
        for i in range(args.warmup):              # every region starts from a primed pipeline
            step(it)
            it += 1
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        last["timing"] = True
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(it)
            it += 1
        drain()
        if trace is not None:
            trace.append(("drain", 0, time.perf_counter(), 0.0))
        torch.cuda.synchronize()
        if trace is not None:
            trace.append(("sync", 0, time.perf_counter(), 0.0))
        if world > 1:
            dist.barrier()
        samples.append(time.perf_counter() - t0)
        last["timing"] = False
    if world > 1:                                 # per region: the slowest rank
        t = torch.tensor(samples, dtype=torch.float64, device=gather.cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        samples = [float(x) for x in t.cpu().tolist()]
    dt = float(np.median(samples))
    last_step = it - 1
    if trace is not None:
        for kind, i, t, ms in trace:
            if t >= t0:
                sys.stderr.write("%-5s %3d %8.1f us  scan %.1f us\n" % (kind, i, (t - t0) * 1e6, ms * 1e3))
        sys.stderr.write("total %.1f us\n" % (dt * 1e6))
    st = last["st"]

    # outside the timed region: what rank 0 holds after the last gather must be every
    # rank's own records of that step (CRC of the 32-byte rows)
    gather_verified = None
    if world > 1:
        import zlib
        last_corpus = corpora[last_step % rotate]
        own, _ = ix.search(last_corpus)
        crc = torch.tensor([zlib.crc32(own.tobytes()), len(own)], dtype=torch.int64, device=gather.cdev)
        crcs = torch.zeros(2 * world, dtype=torch.int64, device=gather.cdev)
        dist.all_gather_into_tensor(crcs, crc)
        holder = gather.roots[last["buf"]]          # the rank that received the last step's records
        ok = 1
        if rank == holder:
            parts, cnts = gather.rows(last["buf"])
            crcs = crcs.cpu().tolist()
            ok = int(all(cnts[r] == crcs[2 * r + 1] and
                         zlib.crc32(parts[r].tobytes()) == crcs[2 * r] for r in range(world)))
        flag = torch.tensor([ok], dtype=torch.int64, device=gather.cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_verified = bool(int(flag.item()))

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_works * args.steps / dt
        if not scan_ms:                       # no timed launch fell into the region
            ix.set_scan_timing(1)
            scan_ms.append(ix.search_end(ix.search_begin(corpora[0], gather.bufs[0].data_ptr(), cap,
                                                         packed=packed, header=True))[1].scan_ms)
        kernel_ms = float(np.mean(scan_ms))
        # the same kernel with nothing else on the GPU: one search at a time
        ix.set_scan_timing(1)
        alone = []
        for i in range(12):
            alone.append(ix.search_end(ix.search_begin(corpora[i % rotate], gather.bufs[0].data_ptr(), cap,
                                                       packed=packed, header=True))[1].scan_ms)
        alone_ms = float(np.mean(alone[2:]))
        kernel = ix.kernel_name(corpora[0])
        # ... and a whole search alone: an index with one lane puts the records into place
        # inside the same launch (no k_compact behind it), one search at a time.  This is the
        # figure `frac` is made of: what rocprofv3 reports for the kernel under --lanes 1
        # (profiles/*_lanes1_kernel_stats.csv)
        search_alone_ms, kernel1 = alone_ms, kernel
        if int(os.environ.get("FS_LANES", "1")) > 1:
            os.environ["FS_LANES"] = "1"
            ix1 = ScriptIndex(script, swords, emb, normals, cfg=cfg)
            os.environ["FS_LANES"] = str(max(1, args.lanes))
            c1s = [ix1.corpus(t, o, chars, coff) for t, o in zip(toks, offs)]
            ix1.set_scan_timing(4)     # (every 4th launch carries the events: an event record costs stream time)
            one, tk = [], []
            # one lane = one stream: the searches run strictly one after the other, each with the
            # GPU to itself; three are kept queued so that the GPU does not fall idle (and drop
            # its clock) between them while the host collects the previous one
            for i in range(80):
                tk.append(ix1.search_begin(c1s[i % rotate], gather.bufs[i % len(gather.bufs)].data_ptr(), cap,
                                           packed=packed, header=True))
                if len(tk) >= min(3, len(gather.bufs)):
                    one.append(ix1.search_end(tk.pop(0))[1].scan_ms)
            while tk:
                one.append(ix1.search_end(tk.pop(0))[1].scan_ms)
            one = [x for x in one[8:] if x > 0]
            search_alone_ms = float(np.mean(one))
            kernel1 = ix1.kernel_name(c1s[0])
            ix1.close()
        # what the launch shape costs for this many bytes with no work at all: a kernel that only
        # reads the ids the way k_scan_rows does (rotating over the distinct batches: from HBM)
        floor_ms = float(np.mean([ix.stream_floor(corpora[i % rotate], reps=4) for i in range(2 * rotate)]))
        rows_step = float(np.mean(rows_per_corpus))
        exact = st.path == abi.FS_MODE_EXACT
        fused = kernel.startswith("k_scan_rows")
        if exact:
            # 4 B per fan token read + 32 B per emitted record.  k_scan_rows
            # is the whole search (ids in, records out); the k_scan8 chain's first kernel
            # only reads the ids.
            algo_bytes = 4.0 * n_tok + (32.0 * rows_step if fused else 0.0)
            note = "4 B per fan token" + (" + 32 B per record" if fused else "")
        elif kernel.startswith("k_scan_near"):
            # the integer prefilter of the LSH pipeline is the kernel the events bracket; the
            # LSH work proper (keys, buckets, distances of the flagged windows) is k_lsh_verify
            algo_bytes = 4.0 * n_tok
            note = "4 B per fan token (prefilter only; the step is dominated by k_lsh_verify)"
        else:
            algo_bytes = float(st.windows_processed) * args.window * 212 * 4
            note = "n rows of 212 float32 projections (848 B) per window (cache-served gather)"
        step_bytes = 4.0 * n_tok + 32.0 * rows_step
        achieved = algo_bytes / (search_alone_ms * 1e-3) / 1e9
        resident = rotate * shard_bytes <= MALL_BYTES
        traffic, traffic_src = None, None
        build = _lib.source_hash()
        tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                # the PMC passes are separate runs (gpurun refuses counters next to a trace): the
                # figure is taken only when it was measured on these very sources and this workload
                if (rec.get("build") == build and rec.get("kernel") == kernel and rec.get("n_tok") == n_tok
                        and rec.get("rotate") == rotate
                        and rec.get("lanes") in (None, int(os.environ.get("FS_LANES", "1")))):
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/scan_traffic.json (build %s): %s" % (build, rec.get("source", "rocprofv3 --pmc"))
                else:
                    traffic_src = ("profiles/scan_traffic.json does not apply (its build %s, this build %s)"
                                   % (rec.get("build"), build))
            except Exception:
                traffic = None
        out = {
            "metric": "fanworks/sec (6-gram search)",
            "value": value,
            "unit": "fanworks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_min": min(samples) / args.steps * 1e3,
            "ms_per_step_max": max(samples) / args.steps * 1e3,
            "samples_ms": [round(x * 1e3, 5) for x in samples],
            "timed_regions": "%d regions of %d steps, each behind %d warm-up steps; median region reported"
                             % (len(samples), args.steps, args.warmup),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            # works differ in length between the workloads (c2: 2000 tokens, c3: 5000), so the
            # rate in tokens is what compares an N = 1 line (c2) with an N > 1 line (c3)
            "tokens_per_s": total_works * tpw * args.steps / dt,
            "config": {"workload": ("%s: %d works x %d tokens split over %d GPU(s) (%d works, %.0f MB of ids "
                                    "per GPU) vs %d-token script, %d-gram"
                                    % (wl, total_works, tpw, world, n_works, shard_bytes / 1e6,
                                       conf["script_tokens"], args.window)) if strong else
                                   ("%s: %d works x %d tokens per GPU vs %d-token script, %d-gram"
                                    % (wl, n_works, tpw, conf["script_tokens"], args.window)),
                       "works_per_gpu": n_works, "tokens_per_work": tpw,
                       "script_tokens": conf["script_tokens"], "window": args.window,
                       "distinct_batches": rotate,
                       "lanes": int(os.environ.get("FS_LANES", "1")),
                       "unique_filter": bool(cfg.unique_filter),
                       "ids_bytes_rotated": rotate * shard_bytes,
                       "rows_per_gpu_step": int(round(rows_step)),
                       "wire_record_bytes": rec_bytes if world > 1 else None,
                       "gather_verified": gather_verified,
                       "pipeline": "%d searches in flight (fs_search_corpus_begin/_end, %d record "
                                   "buffers) on %s lane(s) = stream(s) of the library"
                                   % (inflight, NB, os.environ.get("FS_LANES", "1")),
                       "gather": ("%s %s to rank %s, overlapped" % ("gloo (rehearsal)" if rehearsal else "rccl",
                                                                "all_to_all per N steps," if exchange else "gather",
                                                                "(step mod N)" if any_root else "0"))
                       if world > 1 else "none",
                       "path": ("exact-ngram-scan (synthetic table only: the exact-n-gram proof holds, "
                                "c_max %.3f); tables with near-synonyms take the LSH pipeline, see "
                                "companions.lsh_clustered_table" % ix.info["c_max"]) if exact else "lsh"},
            "roofline": {"bound": "infinity-cache" if resident else "hbm",
                         "kernel": kernel, "bytes_model": note,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algo_bytes, "build": build,
                         "search_ms_alone": search_alone_ms, "kernel_alone": kernel1,
                         "stream_floor_ms": floor_ms,
                         "stream_floor_frac": 4.0 * n_tok / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "stream_floor_note": "k_stream_floor: the ids of a batch read in k_scan_rows' launch shape "
                                              "and nothing else (fs_stream_floor), one launch at a time: the part of "
                                              "search_ms_alone that is the launch and the HBM stream",
                         "launch_ms_alone": alone_ms,
                         "frac_alone": algo_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "overlapped_launch_ms": kernel_ms, "timed_launches": len(scan_ms),
                         "frac_overlapped": algo_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "step": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "step_bytes": step_bytes,
                         "note": "achieved / frac: a whole search by itself (search_ms_alone: an index with one "
                                 "lane = one stream, the one launch of `kernel_alone` takes ids in and puts the "
                                 "records into place, HIP events on its dispatch; the launches run one after the "
                                 "other, three queued, nothing else on the GPU).  launch_ms_alone: "
                                 "the kernel in the shape the timed region launches it in (%d lanes: k_compact "
                                 "puts the records into place behind it), also by itself; overlapped_launch_ms: "
                                 "its dispatch-to-completion time inside the timed region, where %d searches "
                                 "share the GPU; step = (4 B x tokens + 32 B x records) / ms_per_step"
                                 % (int(os.environ.get("FS_LANES", "1")), inflight)},
        }
        if not exact:
            # a search of several launches: the whole search, and the kernel with the largest share
            rec, _, _ = search_companion(ix, corpora[0], n_works, n_tok, inflight)
            out["roofline"] = dict(rec["roofline"], kernels_us=rec["kernels_us"], build=build, traffic=None,
                                   overlapped_first_kernel_ms=kernel_ms)
        if world == 1 and not args.no_companions:
            out["companions"] = companions(ix, corpora, toks, offs, chars, coff, words, script,
                                           swords, emb, args.window, n_works)
            if wl == "c2" and args.window == 6 and not args.works:
                for c in corpora:
                    c.close()
                out["companions"].update(c3_companions(ix, script, chars, coff, inflight))
                out["companions"]["command_end_to_end"] = command_companion()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, script, swords, words, emb, normals, tpw,
                                               chars, coff, args.cpu_seconds)
            if ref_shaped is not None:
                out["cpu_baseline"]["reference_shaped"] = ref_shaped
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
