"""The N>1 path on CPU: gloo processes (world sizes 2 and 4) run analyze() with an
oracle-backed searcher through the product's gather (dist.search_sharded: one
self-describing payload per rank -- records, work offsets, fan words -- a size/failure
agreement and one padded gather); the CSVs must be byte-identical to a single-process
run.  Ranks whose shards differ in record format or are empty, and a rank that fails,
are covered too.  Also the range splitter.  With a GPU
(-m gpu): two ranks with the real AnnIndexSearch, 8-byte wire records left in HBM by
the search and expanded on rank 0."""

import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from fandom_search_amd import abi, dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_contiguous():
    assert dist.split_contiguous([1] * 10, 2) == [0, 5, 10]
    assert dist.split_contiguous([1] * 10, 4) == [0, 3, 5, 8, 10]
    b = dist.split_contiguous([100, 1, 1, 1, 1, 100], 3)
    assert b[0] == 0 and b[-1] == 6 and b == sorted(b)
    assert dist.split_contiguous([], 3) == [0, 0, 0, 0]
    assert dist.split_contiguous([5], 4) == [0, 0, 0, 0, 1] or dist.split_contiguous([5], 4)[-1] == 1
    for parts in (1, 2, 3, 8):
        w = np.random.default_rng(parts).integers(1, 50, size=37)
        b = dist.split_contiguous(w, parts)
        assert len(b) == parts + 1 and b[0] == 0 and b[-1] == 37 and b == sorted(b)


def test_search_sharded_single_process_is_identity():
    """One rank: the shard goes through pack_shard / unpack_shard and comes back unchanged."""
    rows = np.zeros(3, dtype=abi.ROW_DTYPE)
    rows["work"] = [0, 1, 2]
    rows["fan_ix"] = [5, 6, 7]

    class One(object):
        def search_rows(self, filenames):
            return rows, ["a", "b c", "d"]

    out, words = dist.search_sharded(["x", "y", "z"], [1, 1, 1], One())
    assert out.tobytes() == rows.tobytes() and words == ["a", "b c", "d"]


WORKER = textwrap.dedent('''
    import os, sys, types
    sys.path.insert(0, %(root)r)
    import numpy as np
    from fandom_search_amd import abi, search, synth, vocab
    from tests import util

    class OracleSearcher(object):
        """Stands in for AnnIndexSearch on a machine without a GPU: same
        search_rows contract, rows from the plain-C oracle."""
        def __init__(self, script_path):
            rows = search.load_markup_script(script_path)[1:]
            self.word_lowercase = tuple(r[0] for r in rows)
            self.orth_id = tuple(r[1] for r in rows)
            self.scene = tuple(r[2] for r in rows)
            self.character = tuple(r[3] for r in rows)
            self.words = synth.vocab_words()
            self.ids = {w: i for i, w in enumerate(self.words)}
            self.emb = synth.embedding()
            script = np.array([self.ids[w] for w in self.word_lowercase], np.uint32)
            self.cfg = abi.make_config()
            self.oi = util.oracle_index(self.cfg, script, self.words, self.emb,
                                        synth.lsh_normals(6), threads=2)
            self.chars, self.coff = vocab.pack_strings(self.words)
        def search_rows(self, filenames):
            texts = [search.read_work_tokens(f) for f in filenames]
            off = np.zeros(len(texts) + 1, np.uint64)
            off[1:] = np.cumsum([len(t) for t in texts])
            tok = np.array([self.ids[t] for ts in texts for t in ts], np.uint32)
            rows, _ = self.oi.search(tok, off, self.chars, self.coff)
            pos = off[rows["work"]].astype(np.int64) + rows["fan_ix"].astype(np.int64)
            return rows, [self.words[t] for t in tok[pos].tolist()]

    out, me = sys.argv[1], int(os.environ.get("RANK", "0"))
    if os.environ.get("PER_RANK_CWD") and me > 0:      # ranks that do not share a directory
        out = os.path.join(out, "rank%%d" %% me)
        os.makedirs(out, exist_ok=True)
    os.chdir(out)
    if os.environ.get("STALE_BATCH") and me == 0:      # left behind by an earlier run
        with open("match-6gram-batch-1.csv", "w") as fh:
            fh.write("stale" + chr(10))
        os.utime("match-6gram-batch-1.csv", (1, 1))
    if os.environ.get("SCRAMBLE_LISTING") and me > 0:  # a file system that lists for this rank in another order
        real_listdir = os.listdir
        os.listdir = lambda d=".": list(reversed(real_listdir(d)))
    if os.environ.get("FAIL_WRITE_RANK") == str(me):
        def full(records, name):
            raise OSError("no space left on rank %%d" %% me)
        search.write_records = full
    args = types.SimpleNamespace(fan_works=sys.argv[2], script=sys.argv[3],
                                 skip_works=0, num_works=-1)
    search.analyze(args, chunk_size=7, searcher=OracleSearcher(sys.argv[3]))
''')


def _free_port():
    """A rendezvous port nobody holds right now (two test runs on one box must not collide
    on a hard-coded one)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_path, outdir, fandir, script, world, check=True, **extra):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER % dict(root=ROOT))
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, OMP_NUM_THREADS="2", **extra)
    if world == 1:
        cmd = [sys.executable, str(worker), outdir, fandir, script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(worker), outdir, fandir, script]
    r = subprocess.run(cmd, check=check, env=env, timeout=600, cwd=ROOT,
                       stdout=None if check else subprocess.PIPE, stderr=None if check else subprocess.STDOUT, text=True)
    if not check:
        return r
    return {f: open(os.path.join(outdir, f), "rb").read() for f in sorted(os.listdir(outdir))
            if os.path.isfile(os.path.join(outdir, f))}


@pytest.mark.timeout(900)
def test_gloo_runs_write_identical_csvs(tmp_path):
    from fandom_search_amd import synth
    words = synth.vocab_words()
    script = synth.script_tokens(1500)
    fandir = str(tmp_path / "fan")
    # ragged works, including an empty file and one shorter than a window
    lens = [300, 0, 120, 4, 260, 310, 90, 200, 150, 333, 70, 128, 256, 64, 180, 222, 199]
    os.makedirs(fandir)
    for i, n in enumerate(lens):
        tok = synth.fanwork_tokens(i, n, script) if n else []
        with open(os.path.join(fandir, synth.work_name(i)), "w") as fh:
            fh.write(" ".join(words[int(t)] for t in tok))
    spath = str(tmp_path / "script.txt")
    with open(spath, "w") as fh:
        fh.write(synth.script_markup(script, words))
    one = _run(tmp_path, str(tmp_path / "out1"), fandir, spath, 1)
    assert len(one) == 4 and sum(len(v) for v in one.values()) > 1000   # 3 batch files + dated file
    for world in (2, 4):
        many = _run(tmp_path, str(tmp_path / ("out%d" % world)), fandir, spath, world)
        assert list(one) == list(many)
        for name in one:
            assert one[name] == many[name], (world, name)


@pytest.mark.timeout(900)
def test_listing_order_os_is_rank_zeros_on_every_rank(tmp_path):
    """--listing os / FANDOM_SEARCH_LISTING=os: the file system's own listing goes into the
    seeded shuffle, and under a launcher every rank works on rank 0's list -- also when its own
    listdir() answers in another order (here: reversed for the ranks > 0)."""
    fandir, spath = _small_corpus(tmp_path)
    one = _run(tmp_path, str(tmp_path / "out1"), fandir, spath, 1, FANDOM_SEARCH_LISTING="os")
    two = _run(tmp_path, str(tmp_path / "out2"), fandir, spath, 2, FANDOM_SEARCH_LISTING="os", SCRAMBLE_LISTING="1")
    assert list(one) == list(two) and len(one) == 4
    for name in one:
        assert one[name] == two[name], name
    # (and it is the listing's order that went in: the sorted default gives other batches
    # unless the file system happens to list in sorted order)
    import random
    from fandom_search_amd import search
    names = os.listdir(fandir)
    want = list(names)
    random.seed(search.SHUFFLE_SEED)
    random.shuffle(want)
    first = one["match-6gram-batch-0.csv"].split(b"\r\n")[0].split(b",")[0].decode()
    assert os.path.basename(first) in want[:7]


def _small_corpus(tmp_path):
    from fandom_search_amd import synth
    words = synth.vocab_words()
    script = synth.script_tokens(1500)
    fandir = str(tmp_path / "fan")
    os.makedirs(fandir)
    for i, n in enumerate([300, 120, 260, 310, 90, 200, 150, 333, 70, 128, 256, 64, 180, 222, 199]):
        with open(os.path.join(fandir, synth.work_name(i)), "w") as fh:
            fh.write(" ".join(words[int(t)] for t in synth.fanwork_tokens(i, n, script)))
    spath = str(tmp_path / "script.txt")
    with open(spath, "w") as fh:
        fh.write(synth.script_markup(script, words))
    return fandir, spath


@pytest.mark.timeout(900)
def test_ranks_without_a_shared_directory_and_a_stale_batch_file(tmp_path):
    """the ranks take turns at writing batch files and rank 0 concatenates them.
    Ranks with a working directory of their own (several nodes), and a batch file an earlier
    run left in rank 0's directory: rank 0 notices (size and age against what the writer
    reports) and has the bytes sent over; its directory ends up as a one-rank run's."""
    fandir, spath = _small_corpus(tmp_path)
    one = _run(tmp_path, str(tmp_path / "out1"), fandir, spath, 1)
    assert len(one) == 4
    two = _run(tmp_path, str(tmp_path / "out2"), fandir, spath, 2, PER_RANK_CWD="1", STALE_BATCH="1")
    assert list(one) == list(two)
    for name in one:
        assert one[name] == two[name], name


@pytest.mark.timeout(600)
def test_a_rank_that_cannot_write_its_batch_file_stops_every_rank(tmp_path):
    """the writer of a batch fails on one rank (disk full): the others must not wait
    in a barrier for it -- every rank ends, the failing one with its own exception."""
    import time
    fandir, spath = _small_corpus(tmp_path)
    t0 = time.time()
    r = _run(tmp_path, str(tmp_path / "outf"), fandir, spath, 2, check=False, FAIL_WRITE_RANK="1")
    assert r.returncode != 0
    assert time.time() - t0 < 120, "a rank waited for the failed one"
    assert "no space left on rank 1" in r.stdout
    assert "RankFailed" in r.stdout


MIXED_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch
    import torch.distributed as tdist
    from fandom_search_amd import abi, dist

    rank, _, world = dist.init_from_env("gloo")
    mode = sys.argv[1]

    class Searcher(object):
        """Rank 0: fs_row records (as from a shard with an out-of-vocabulary token);
        the last rank: an EMPTY shard that announces 8-byte wire records; `fail`: the
        last rank raises instead."""
        engine = None
        def search_shard(self, sub):
            if rank == world - 1:
                if mode == "fail":
                    raise ValueError("boom on rank %%d" %% rank)
                return dist.Shard(torch.zeros(dist.HDR + 8, dtype=torch.uint8), 8, 0,
                                  np.zeros(len(sub) + 1, np.uint64), [])
            rows = np.zeros(len(sub), dtype=abi.ROW_DTYPE)
            rows["work"] = np.arange(len(sub))
            rows["fan_ix"] = 7 + rank
            rows["comb"] = 0.5
            return dist.shard_from_rows(rows, ["w%%d_%%d" %% (rank, i) for i in range(len(sub))], len(sub))

    files = ["f%%d" %% i for i in range(9)]
    try:
        rows, words = dist.search_sharded(files, [1] * 9, Searcher())
    except dist.RankFailed:
        assert mode == "fail" and rank != world - 1
        sys.exit(3)
    except ValueError as e:
        assert mode == "fail" and rank == world - 1
        print("raised:", e, flush=True)
        sys.exit(4)
    assert mode == "mixed"
    if rank == 0:
        b = dist.split_contiguous([1] * 9, world)
        n = b[world - 1]
        assert len(rows) == n and rows["work"].tolist() == list(range(n)), rows["work"]
        assert words[0] == "w0_0" and len(words) == n
        print("MIXED_OK", n, flush=True)
    else:
        assert rows is None and words is None
    dist.finalize()
''')


def _torchrun(worker, world, port, *argv):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                           "--master-port", str(port), str(worker)] + list(argv),
                          env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=300, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.timeout(600)
def test_ranks_with_different_record_formats_and_an_empty_shard(tmp_path):
    """one rank holds fs_row records, another an empty shard of 8-byte
    wire records.  The payloads describe themselves, so the gather neither hangs nor
    decodes one rank's records with another's format."""
    worker = tmp_path / "mixed.py"
    worker.write_text(MIXED_WORKER % dict(root=ROOT))
    for world in (2, 3):
        r = _torchrun(worker, world, _free_port(), "mixed")
        assert r.returncode == 0, r.stdout[-3000:]
        assert "MIXED_OK" in r.stdout


@pytest.mark.timeout(600)
def test_a_failing_rank_stops_every_rank(tmp_path):
    """A rank that raises in its share tells the others through the size agreement:
    nobody is left waiting in the gather (the run ends at once, non-zero)."""
    import time
    worker = tmp_path / "mixed.py"
    worker.write_text(MIXED_WORKER % dict(root=ROOT))
    t0 = time.time()
    r = _torchrun(worker, 2, _free_port(), "fail")
    assert r.returncode != 0
    assert time.time() - t0 < 120, "the surviving rank waited for the failed one"
    assert "boom on rank 1" in r.stdout


GPU_WORKER = textwrap.dedent('''
    import os, sys, types
    sys.path.insert(0, %(root)r)
    from fandom_search_amd import search
    os.chdir(sys.argv[1])
    args = types.SimpleNamespace(fan_works=sys.argv[2], script=sys.argv[3],
                                 skip_works=0, num_works=-1)
    search.analyze(args, chunk_size=9)
''')


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_with_the_hip_searcher(tmp_path):
    """`ao3.py search` under torch.distributed.run with the real AnnIndexSearch: every
    rank searches its share on the GPU, the 8-byte wire records go from HBM into the
    gather and are expanded on rank 0.  (One GPU here, so the ranks share it and the
    collective is gloo: FANDOM_SEARCH_DIST_BACKEND; with one GPU per rank the same code
    runs over RCCL.)  Byte-identical to the single-process run."""
    from fandom_search_amd import synth
    words = synth.vocab_words()
    script = synth.script_tokens(1500)
    fandir = str(tmp_path / "fan")
    lens = [300, 0, 120, 4, 260, 310, 90, 200, 150, 333, 70, 128, 256, 64, 180, 222, 199, 6, 5]
    os.makedirs(fandir)
    for i, n in enumerate(lens):
        tok = synth.fanwork_tokens(i, n, script) if n else []
        with open(os.path.join(fandir, synth.work_name(i)), "w") as fh:
            fh.write(" ".join(words[int(t)] for t in tok))
    spath = str(tmp_path / "script.txt")
    with open(spath, "w") as fh:
        fh.write(synth.script_markup(script, words))
    worker = tmp_path / "gpu_worker.py"
    worker.write_text(GPU_WORKER % dict(root=ROOT))
    outs = {}
    for world in (1, 2):
        outdir = str(tmp_path / ("gout%d" % world))
        os.makedirs(outdir)
        env = dict(os.environ, FANDOM_SEARCH_DIST_BACKEND="gloo", FANDOM_SEARCH_SYNTHETIC_VOCAB="1")
        if world == 1:
            cmd = [sys.executable, str(worker), outdir, fandir, spath]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                   "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), str(worker), outdir, fandir, spath]
        subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
        outs[world] = {f: open(os.path.join(outdir, f), "rb").read() for f in sorted(os.listdir(outdir))}
    assert list(outs[1]) == list(outs[2]) and len(outs[1]) == 4
    for name in outs[1]:
        assert outs[1][name] == outs[2][name], name
    assert sum(len(v) for v in outs[1].values()) > 1000


def test_two_ranks_on_one_device_are_refused_up_front():
    """The first RCCL run must not end in a hang inside a collective: fewer visible devices
    than ranks, or two ranks with the same device identity, raise DeviceMapError before any
    communicator is formed (dist.init_nccl_checked; the identities travel through the
    rendezvous store, here an in-process one driven from two threads)."""
    import threading
    import torch.distributed as tdist
    from fandom_search_amd import dist as fdist
    fdist.check_device_count(1, 2, 2)
    fdist.check_device_count(7, 8, 8)
    with pytest.raises(fdist.DeviceMapError, match="2 ranks on this node but 1 GPU"):
        fdist.check_device_count(1, 2, 1)
    with pytest.raises(fdist.DeviceMapError):
        fdist.check_device_count(0, 8, 4)

    def run(idents):
        store = tdist.HashStore()
        errs = [None] * len(idents)

        def rank(r):
            try:
                fdist.verify_distinct_devices(store, r, len(idents), idents[r], timeout_s=20)
            except Exception as e:
                errs[r] = e

        ts = [threading.Thread(target=rank, args=(r,)) for r in range(len(idents))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(30)
        return errs

    assert run(["box/0:1:0/a", "box/0:2:0/b", "box/0:3:0/c"]) == [None, None, None]
    errs = run(["box/0:1:0/a", "box/0:1:0/a"])
    assert all(isinstance(e, fdist.DeviceMapError) for e in errs)
    assert "ranks 0 and 1 both map to device box/0:1:0/a" in str(errs[0])


def test_bench_refuses_an_nccl_run_with_fewer_devices_than_ranks():
    """`bench.py --gpus 2 --backend nccl` on a box with fewer than two GPUs (this container:
    none) ends at once with the reason, in the parent, before any rank is started."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl",
                        "--steps", "1", "--warmup", "0"], cwd=ROOT, timeout=300, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices here: the run itself is the driver's to make")
    assert r.returncode != 0 and "GPU(s) visible" in r.stdout and "gloo" in r.stdout
