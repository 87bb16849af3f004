#!/usr/bin/env python3
"""fanworks/sec of the 6-gram search hot path on MI355X.

A step = one pass of the search over one batch of synthetic fan works that is
already resident in HBM (fs_search_corpus of include/fandom_search.h: scan
kernel + verify + Levenshtein + per-word records, rows left in HBM).  At N=1
the batch is BASELINE.json configs[1] ("c2": 10k works x 2k tokens vs a
2k-line script, 6-gram); with N ranks every rank searches its own c2-sized
shard of distinct works (weak scaling) and the match rows of all ranks are
gathered to rank 0 over RCCL inside the step (double-buffered, so the gather
of step i overlaps the scan of step i+1).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     scan kernel: algorithmic bytes (4 B per fan token)
               / HIP-event duration of the kernel, against 8 TB/s HBM
  cpu_baseline the plain-C oracle (the reference's LSH algorithm) on a bounded
               sample of the same workload on the host cores (N=1 only), and as
               `reference_shaped` the literal Python restatement on configs[0] in
               a 4-process pool, the way search.py:381-385 runs the original
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2",
                    help="per-rank shard: c2 (default), c3shard, c3, c1")
    ap.add_argument("--works", type=int, default=0, help="override works per rank")
    ap.add_argument("--window", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl",
                    help="nccl (RCCL, one GPU per rank) or gloo (rehearsal: every rank "
                         "computes on GPU 0, rows gathered through host memory)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--lanes", type=int, default=0,
                    help="streams the library spreads searches over (FS_LANES; library "
                         "default 1): 2 or 4 trade the scan kernel's own speed for step rate")
    ap.add_argument("--wire", type=int, default=0, help="N > 1: force 16-byte wire records")
    ap.add_argument("--inflight", type=int, default=2,
                    help="searches kept in flight (the library overlaps them on its lanes)")
    ap.add_argument("--no-reference-shaped", action="store_true",
                    help="skip the Python reference-shaped leg of cpu_baseline")
    return ap.parse_args()


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
    from fandom_search_amd import synth, vocab
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
                                    arith=nr.LiteralArith(), unique_filter=True)
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


def main():
    args = parse()
    if args.lanes:
        os.environ["FS_LANES"] = str(args.lanes)
    ref_shaped = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline
            and not args.no_reference_shaped):
        ref_shaped = cpu_reference_shaped(args.window)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    import torch
    import torch.distributed as dist
    from fandom_search_amd import _lib, abi, synth, vocab
    from fandom_search_amd.engine import ScriptIndex

    rehearsal = args.backend == "gloo"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cdev = "cpu" if rehearsal else "cuda"        # where collectives run

    conf = dict(synth.CONFIGS[args.workload])
    if args.works:
        conf["n_works"] = args.works
    n_works, tpw = conf["n_works"], conf["tokens_per_work"]
    words = synth.vocab_words()
    emb = synth.embedding()
    normals = synth.lsh_normals(args.window)
    script = synth.script_tokens(conf["script_tokens"])
    swords = [words[int(t)] for t in script]
    chars, coff = vocab.pack_strings(words)
    tok, off = synth.corpus_tokens(n_works, tpw, script, first_work=rank * n_works)

    cfg = abi.make_config(window_size=args.window, device=dev_index)
    ix = ScriptIndex(script, swords, emb, normals, cfg=cfg)
    corpus = ix.corpus(tok, off, chars, coff)

    # row buffers in HBM.  With more than one rank the exact pipeline emits wire
    # records for the gather, which rank 0 expands again without loss: 8 bytes each
    # (token position + packed script position / offset / Levenshtein; rank 0 needs the
    # ranks' work offsets for them, gathered once below), or 16 bytes for scripts of
    # 2^18 tokens and more.
    packed = False
    if world > 1 and ix.info["path"] == abi.FS_MODE_EXACT:
        packed = 8 if len(script) < abi.PACKED8_MAX_SCRIPT and args.wire != 16 else 16
    rec_bytes = packed if packed else 32
    # every row buffer starts with a 32-byte header whose first 8 bytes carry the
    # row count of the step: count and records travel in ONE gather per step
    HDR = 32
    cap = max(4096, corpus.n_tok // 16)

    def row_buffer():
        return torch.zeros(HDR + cap * rec_bytes, dtype=torch.uint8, device="cuda")

    while True:
        bufs = [row_buffer() for _ in range(2)]
        try:
            n_rows, st = ix.search_device(corpus, bufs[0].data_ptr() + HDR, cap, packed=packed)
            break
        except _lib.FsError as e:
            if e.code != abi.FS_E_CAPACITY:
                raise
            cap = int(e.required * 1.05) + 64
    pad = cap
    if world > 1:
        t = torch.tensor([cap], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pad = int(t.item())
        if pad != cap:
            cap = pad
            bufs = [row_buffer() for _ in range(2)]
        gathered = full_rows = None
        if rank == 0:
            # one contiguous landing buffer per slot; rank r's header + records at
            # [r * (HDR + cap * rec_bytes), ...)
            gathered = [torch.zeros(world * (HDR + cap * rec_bytes), dtype=torch.uint8,
                                    device=cdev) for _ in range(2)]
            if packed:
                full_rows = torch.empty(world * cap * 32, dtype=torch.uint8, device="cuda")
        all_off = None
        if packed == 8:
            # the batch layout of every rank, once per corpus: (n_works + 1) offsets
            mine_off = torch.from_numpy(off.astype(np.int64)).to(cdev)
            all_off = torch.zeros(world * len(off), dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(all_off, mine_off)
            all_off = all_off.cuda() if rank == 0 else None

    # Software pipeline over NB row buffers: the search of step i is queued while
    # the GPU still finishes step i-1 (fs_search_corpus_begin / _end), and with more
    # than one rank the gather of step i-1 runs beside the search of step i.
    NB = args.inflight + 1
    while len(bufs) < NB:
        bufs.append(row_buffer())
    if world > 1 and rank == 0:
        while len(gathered) < NB:
            gathered.append(torch.zeros(world * (HDR + cap * rec_bytes), dtype=torch.uint8,
                                        device=cdev))
    pending = [None] * NB       # gathers in flight, per buffer
    tickets = {}                # step -> (ticket, buffer)
    scan_ms = []
    total_ms = []
    total_rows = 0
    last_st = [None]
    last_gathered = [0]

    def finish(b):
        """Buffer b is about to be reused: its gather must be complete.  Work.wait()
        on RCCL only orders torch's current stream, so the host also waits for that
        stream (the library writes the buffers from its own stream).  Rank 0 now
        holds every rank's records of that step: 32-byte rows, or the lossless
        16-byte wire records, which are expanded where they are consumed
        (fs_rows_unpack; done once after the timed region for the verification)."""
        if pending[b] is None:
            return
        for h in pending[b]:
            h.wait()
        pending[b] = None
        if not rehearsal:
            torch.cuda.current_stream().synchronize()
        last_gathered[0] = b

    # the scan kernel is timed (events attached to its dispatch) on every 4th step
    # (a timed search carries extra event records, about 15 us: time few of them)
    ix.set_scan_timing(4 if args.steps >= 8 else 2 if args.steps >= 4 else 1)

    def complete(i):
        """Finish the search of step i and hand its rows to the gather."""
        nonlocal total_rows
        t, b = tickets.pop(i)
        n, st = ix.search_end(t)
        if st.scan_ms > 0:
            scan_ms.append(st.scan_ms)
            total_ms.append(st.total_ms)
        total_rows = n
        last_st[0] = st
        if world > 1:
            send = bufs[b].cpu() if rehearsal else bufs[b]
            h = dist.gather(send, list(gathered[b].chunk(world)) if rank == 0 else None, dst=0,
                            async_op=True)
            pending[b] = (h,)

    def step(i):
        b = i % NB
        finish(b)
        # the library writes the step's row count into the buffer's header itself
        tickets[i] = (ix.search_begin(corpus, bufs[b].data_ptr(), cap, packed=packed,
                                      header=True), b)
        if i - (args.inflight - 1) in tickets:
            complete(i - (args.inflight - 1))

    def drain():
        for i in sorted(tickets):
            complete(i)
        for b in range(NB):
            finish(b)

    for i in range(args.warmup):
        step(i)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    scan_ms.clear()
    total_ms.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    drain()
    st = last_st[0]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # outside the timed region: what rank 0 holds after the last gather must be
    # every rank's own records (CRC of the 32-byte rows)
    gather_verified = None
    if world > 1:
        import zlib
        own, _ = ix.search(corpus)
        crc = torch.tensor([zlib.crc32(own.tobytes()), len(own)], dtype=torch.int64, device=cdev)
        crcs = torch.zeros(2 * world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(crcs, crc)
        if rank == 0:
            last = last_gathered[0]
            crcs = crcs.cpu().tolist()
            stride = HDR + cap * rec_bytes
            src = gathered[last].cuda() if rehearsal else gathered[last]
            cnts = src.view(world, stride)[:, :8].contiguous().view(torch.int64).flatten().cpu().tolist()
            if packed:
                for r in range(world):
                    if packed == 8:
                        ix.unpack8_device(src.data_ptr() + r * stride + HDR, min(cnts[r], cap),
                                          all_off.data_ptr() + r * len(off) * 8, len(off) - 1,
                                          full_rows.data_ptr() + r * cap * 32)
                    else:
                        ix.unpack_device(src.data_ptr() + r * stride + HDR, min(cnts[r], cap),
                                         full_rows.data_ptr() + r * cap * 32)
                landed = full_rows.cpu().numpy()
            else:
                landed = src.view(world, stride)[:, HDR:].contiguous().cpu().numpy().reshape(-1)
            gather_verified = True
            for r in range(world):
                chunk = landed[r * cap * 32:(r * cap + cnts[r]) * 32]
                if cnts[r] != crcs[2 * r + 1] or zlib.crc32(chunk.tobytes()) != crcs[2 * r]:
                    gather_verified = False

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * n_works * args.steps / dt
        if not scan_ms:                       # no timed launch fell into the region
            ix.set_scan_timing(1)
            scan_ms.append(ix.search_end(ix.search_begin(corpus, bufs[0].data_ptr(), cap,
                                                         packed=packed, header=True))[1].scan_ms)
        scan_avg_ms = float(np.mean(scan_ms))
        exact = st.path == abi.FS_MODE_EXACT
        if exact:
            # fs_scan_tpl: eight tokens per lane (byte bitmap) while the ids fit 256 MiB
            if corpus.n_tok * 4 <= (256 << 20) and 2 <= args.window <= 8:
                kernel = "k_scan8<%d>" % args.window
            else:
                kernel = "k_scan<%d,4,shuffle,nt>" % args.window
            algo_bytes = 4.0 * corpus.n_tok       # 4 B per fan token
            note = "4 B per fan token"
        else:
            # LSH pipeline: per window n rows of H*B float64 projections are gathered
            # (L2 / Infinity Cache resident table, so 'hbm' is nominal here)
            kernel = "k_lsh_scan"
            algo_bytes = float(st.windows_processed) * args.window * 212 * 4
            note = "n rows of 212 float32 projections (848 B) per window (cache-served gather)"
        achieved = algo_bytes / (scan_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                if rec.get("workload") == args.workload and rec.get("window") == args.window \
                        and rec.get("n_tok") == corpus.n_tok:
                    traffic = rec.get("hbm_bytes_per_launch")
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
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%s: %d works x %d tokens per GPU vs %d-token script, "
                                   "%d-gram" % (args.workload, n_works, tpw,
                                                conf["script_tokens"], args.window),
                       "works_per_gpu": n_works, "tokens_per_work": tpw,
                       "script_tokens": conf["script_tokens"], "window": args.window,
                       "rows_per_gpu_step": int(total_rows),
                       "wire_record_bytes": rec_bytes if world > 1 else None,
                       "gather_verified": gather_verified,
                       "pipeline": "%d searches in flight (fs_search_corpus_begin/_end, %d row "
                                   "buffers) on %s lane(s) = stream(s) of the library"
                                   % (args.inflight, NB, os.environ.get("FS_LANES", "1")),
                       "gather": ("%s gather to rank 0, overlapped" % ("gloo (rehearsal)" if rehearsal
                                                                        else "rccl")) if world > 1 else "none",
                       "path": "exact-ngram-scan" if st.path == abi.FS_MODE_EXACT else "lsh"},
            "roofline": {"bound": "hbm", "kernel": kernel, "bytes_model": note,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "avg_launch_ms": scan_avg_ms, "timed_launches": len(scan_ms)},
            "device_total_ms": float(np.mean(total_ms)) if total_ms and np.mean(total_ms) > 0 else None,
        }
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
