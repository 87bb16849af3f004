"""Multi-GPU execution of the search: one process per GPU, works sharded,
match rows gathered to rank 0.

The reference's only parallelism is a 4-process multiprocessing.Pool over the
works of a batch whose results are concatenated in input order
(/root/reference/search.py:381-386).  Here the works of a batch are split into
contiguous ranges, rank r searches range r on its own GPU against a replicated
script index (built from the same inputs on every rank: no broadcast), and the
variable-length record buffers are gathered to rank 0: the records stay in HBM from
the search kernel to the collective (8-byte wire records of the exact pipeline, one
padded gather: RCCL has no gatherv).  Rank 0 concatenates in rank order == work
order, so the output bytes do not depend on the number of GPUs.
  * `ao3.py search` under torch.distributed.run: search.analyze -> search_sharded,
    one self-describing payload per rank and batch (records + work offsets + fan
    words), an 8-byte size/failure agreement and one gather.
  * bench.py --gpus N: RowGather, fixed device buffers [count header | records]
    filled by the search kernel itself, one gather per step, overlapped.

Backend: "nccl" (= RCCL over xGMI) when a GPU is present, "gloo" otherwise
(CPU tests).
"""

import os

import numpy as np

from . import abi


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class DeviceMapError(RuntimeError):
    """Two ranks of an RCCL run would use the same GPU."""


def check_device_count(local_rank, local_world, visible):
    """Before anything else of a one-GPU-per-rank run: `visible` devices must cover the
    node's ranks.  RCCL refuses a communicator with two ranks on one device only after
    its bootstrap, with a message about duplicate GPUs from inside the first collective
    (or a hang, when only some of the ranks get that far); this says it up front."""
    if visible < local_world or local_rank >= visible:
        raise DeviceMapError(
            "%d ranks on this node but %d GPU(s) visible (LOCAL_RANK %d): two ranks would share a "
            "device and the RCCL communicator cannot be formed.  Run with --gpus <= %d, or, to "
            "rehearse the multi-rank code on one GPU, with the gloo backend (bench.py --backend "
            "gloo / FANDOM_SEARCH_DIST_BACKEND=gloo)" % (local_world, visible, local_rank, max(1, visible)))


def verify_distinct_devices(store, rank, world, ident, timeout_s=60.0):
    """Every rank publishes what tells its GPU apart on its node (host name + PCI bus id, or
    the device's UUID) in the rendezvous store -- plain TCP, no collective -- and reads the
    others': the same identity on two ranks raises DeviceMapError on both, naming them."""
    import datetime
    store.set("fs_device/%d" % rank, ident)
    keys = ["fs_device/%d" % r for r in range(world)]
    store.wait(keys, datetime.timedelta(seconds=timeout_s))
    seen = {}
    for r, k in enumerate(keys):
        v = store.get(k)
        v = v.decode("utf-8", "replace") if isinstance(v, bytes) else str(v)
        if v in seen:
            raise DeviceMapError("ranks %d and %d both map to device %s: one process per GPU is "
                                 "required (check LOCAL_RANK and the *_VISIBLE_DEVICES variables)"
                                 % (seen[v], r, v))
        seen[v] = r


def device_identity(index):
    """Host name + what identifies device `index` physically."""
    import socket
    import torch
    p = torch.cuda.get_device_properties(index)
    uuid = getattr(p, "uuid", None)
    bus = "%s:%s:%s" % (getattr(p, "pci_domain_id", "?"), getattr(p, "pci_bus_id", "?"),
                        getattr(p, "pci_device_id", "?"))
    return "%s/%s/%s" % (socket.gethostname(), bus, uuid)


def init_nccl_checked(local_rank, world):
    """init_process_group("nccl") for one process per GPU, failing fast and clearly when two
    ranks map to one device (the first real RCCL run must not end in a hang): device count
    first, then the process group WITHOUT a bound device (no communicator is formed yet),
    the identities compared through the store, and only then the first collective."""
    import torch
    import torch.distributed as dist
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE") or world)
    check_device_count(local_rank, local_world, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl")
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        verify_distinct_devices(_get_default_store(), dist.get_rank(), world, device_identity(local_rank))
    except DeviceMapError:
        raise
    except Exception as e:              # (a store without wait/get, a torch without the accessor: not fatal)
        import sys
        print("warning: device map not verified (%r)" % (e,), file=sys.stderr)
    dist.barrier(device_ids=[local_rank])       # the communicator, here and now


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when
    WORLD_SIZE > 1.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                # FANDOM_SEARCH_DIST_BACKEND=gloo with GPUs: rehearsal, every rank on GPU 0
                backend = os.environ.get("FANDOM_SEARCH_DIST_BACKEND") or \
                    ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                init_nccl_checked(local_rank, world)
            else:
                dist.init_process_group(backend)
    return rank, local_rank, world


def finalize():
    """Leave together: a rank that exits while others still use the group takes the
    collective backend's threads down mid-flight."""
    import sys
    if "torch.distributed" not in sys.modules:     # never initialised in this process
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def split_contiguous(weights, parts):
    """Boundaries b[0..parts] of contiguous ranges with near-equal weight sums
    (range r = [b[r], b[r+1])).  Deterministic; every item lands in exactly one
    range; ranges may be empty."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if parts <= 0:
        raise ValueError("parts must be positive")
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, parts):
        target = total * r / parts
        cut = int(np.searchsorted(cum, target, side="left"))
        cut = min(max(cut, bounds[-1]), n)
        bounds.append(cut)
    bounds.append(n)
    return bounds


HDR = 32      # bytes in front of the records of a device row buffer (FS_ROWS_HEADER)


class RowGather(object):
    """Gather of device-resident match records to rank 0, one collective per search.

    Every rank owns `n_buffers` device buffers [32-byte header | cap records]; a search
    (ScriptIndex.search_begin(..., header=True)) writes its record count into the
    header and the records behind it, both on the GPU, and `start(b)` hands the whole
    buffer to ONE padded torch.distributed.gather (RCCL over xGMI with the nccl
    backend): count and records travel together, nothing visits the host in between.
    Rank 0 receives rank r's buffer at landing[b][r].  With the exact pipeline the
    records are 8-byte wire records {token position, orig_ix | k << 18 | lev << 22}
    (a quarter of fs_row; 16-byte ones for scripts of 2^18 tokens and more), expanded
    without loss by `unpack` from the ranks' work offsets, which are gathered once per
    corpus.  `rehearsal`: gloo between ranks that all compute on GPU 0 (the buffers
    go through host memory); used to rehearse the N > 1 path on a one-GPU box.

    This is what stands where the reference concatenates the lists its pool workers
    return (/root/reference/search.py:381-386)."""

    def __init__(self, index, cap, rec_bytes, n_buffers=1, group=None, rehearsal=False, any_root=False,
                 exchange=False, inflight=1):
        import torch
        import torch.distributed as dist
        self.index, self.rec_bytes, self.group, self.rehearsal = index, int(rec_bytes), group, rehearsal
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dev = "cuda" if torch.cuda.is_available() else "cpu"     # where the searches write
        self.cdev = "cpu" if rehearsal or self.dev == "cpu" else "cuda"   # where collectives run
        if self.world > 1:                     # one capacity for all ranks (padded gather)
            t = torch.tensor([int(cap)], dtype=torch.int64, device=self.cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            cap = int(t.item())
        self.cap = int(cap)
        self.stride = HDR + self.cap * self.rec_bytes
        # exchange: the buffers come in groups of `world` (buffer b belongs to group b // world and
        # is meant for rank b % world); when a group is full ONE all_to_all_single sends every
        # buffer to its rank.  What arrives is what `world` gathers with the root going round
        # deliver, but every rank uses all its links at once instead of one link per step
        # (2.4 MB per C2 step: 40 us over one xGMI link, 5 us per step over seven).
        self.exchange = bool(exchange) and self.world > 1
        if self.exchange:
            any_root = True
            # a buffer is written again only after its group's all_to_all has been queued: with
            # `inflight` searches queued ahead that takes world + inflight - 1 buffers, whole groups
            need = -(-(self.world + max(1, int(inflight)) - 1) // self.world)
            n_groups = max(2, need, -(-n_buffers // self.world))
            n_buffers = n_groups * self.world
            self.gsend = [torch.zeros(self.world * self.stride, dtype=torch.uint8, device=self.dev)
                          for _ in range(n_groups)]
            self.bufs = [self.gsend[b // self.world][(b % self.world) * self.stride:
                                                     (b % self.world + 1) * self.stride]
                         for b in range(n_buffers)]
            self.gfilled = [set() for _ in range(n_groups)]   # slots filled since the group's last send
            self.gpending = [None] * n_groups
        else:
            self.bufs = [torch.zeros(self.stride, dtype=torch.uint8, device=self.dev)
                         for _ in range(n_buffers)]
        self.n_buffers = n_buffers
        # any_root: start(b, root) may name any rank as the receiver (the bench lets the root go
        # round, step i to rank i mod N: every pair of GPUs has its own xGMI link, and seven
        # links ending at one GPU carry less than the ranks' searches produce)
        self.any_root = bool(any_root)
        self.roots = [0] * n_buffers
        self.landing = None
        if self.exchange:
            # (the buffers of a group share the group's landing area: a rank receives one step of it)
            self.glanding = [torch.zeros(self.world * self.stride, dtype=torch.uint8, device=self.cdev)
                             for _ in range(len(self.gsend))]
            self.landing = [self.glanding[b // self.world] for b in range(n_buffers)]
        elif self.world > 1 and (self.rank == 0 or self.any_root):
            self.landing = [torch.zeros(self.world * self.stride, dtype=torch.uint8, device=self.cdev)
                            for _ in range(n_buffers)]
        self.pending = [None] * n_buffers
        self.all_off = None
        self.off_len = 0

    def set_offsets(self, work_off):
        """Work offsets of this rank's batch (all ranks: same number of works is not
        required).  Needed by rank 0 to expand 8-byte records; one gather per corpus."""
        import torch
        import torch.distributed as dist
        off = np.ascontiguousarray(work_off, dtype=np.int64)
        self.off_len = len(off)
        if self.world > 1:
            n = torch.tensor([len(off)], dtype=torch.int64, device=self.cdev)
            dist.all_reduce(n, op=dist.ReduceOp.MAX, group=self.group)
            self.off_len = int(n.item())
        # [offsets padded with the last one | number of works]
        mine = torch.full((self.off_len + 1,), int(off[-1]) if len(off) else 0, dtype=torch.int64)
        mine[:len(off)] = torch.from_numpy(off)
        mine[self.off_len] = len(off) - 1
        if self.world == 1:
            self.all_off = mine.to(self.dev)
            return
        mine = mine.to(self.cdev)
        out = torch.zeros(self.world * (self.off_len + 1), dtype=torch.int64, device=self.cdev)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        self.all_off = out.to(self.dev) if (self.rank == 0 or self.any_root) else None

    def start(self, b, root=0):
        """Queue the gather of buffer b to rank `root` (after the search that fills it has ended)."""
        import torch.distributed as dist
        if self.world == 1:
            return
        if self.exchange:
            g = b // self.world
            if root != b % self.world:
                raise ValueError("RowGather(exchange=True): buffer %d goes to rank %d, not %d"
                                 % (b, b % self.world, root))
            self.roots[b] = b % self.world
            self.gfilled[g].add(b % self.world)
            if b % self.world == self.world - 1:      # the group's last buffer: steps come in order
                self.flush_group(g)
            return
        if root != 0 and not self.any_root:
            raise ValueError("RowGather(any_root=True) for a root other than rank 0")
        self.roots[b] = root
        send = self.bufs[b].to(self.cdev)           # no copy unless the collective runs elsewhere
        recv = list(self.landing[b].chunk(self.world)) if self.rank == root else None
        self.pending[b] = dist.gather(send, recv, dst=root, group=self.group, async_op=True)

    def flush_group(self, g):
        """exchange: send group g now (also when it is not full: the end of a run; no search
        may be in flight on the group's buffers then).  Slots that hold nothing new -- not
        filled yet, or delivered by an earlier send of this group -- travel with a record
        count of zero, so a receiver never takes a stale step for a new one."""
        import torch.distributed as dist
        if not self.exchange or not self.gfilled[g]:
            return
        for slot in range(self.world):
            if slot not in self.gfilled[g]:
                self.bufs[g * self.world + slot][:8].zero_()
        self.gfilled[g] = set()
        send = self.gsend[g].to(self.cdev)
        self.gpending[g] = dist.all_to_all_single(self.glanding[g], send, group=self.group, async_op=True)

    def flush(self):
        """exchange: send every group that holds unsent buffers."""
        if self.exchange:
            for g in range(len(self.gsend)):
                self.flush_group(g)

    def wait(self, b):
        """Buffer b may be written again / read on its root after this.  Work.wait() on
        RCCL only orders torch's current stream, and the library writes the buffers
        from its own streams, so the host also waits for that stream."""
        import torch
        if self.exchange:
            g = b // self.world
            if self.gpending[g] is not None:
                self.gpending[g].wait()
                self.gpending[g] = None
                if self.cdev == "cuda":
                    torch.cuda.current_stream().synchronize()
            return
        if self.pending[b] is None:
            return
        self.pending[b].wait()
        self.pending[b] = None
        if self.cdev == "cuda":
            torch.cuda.current_stream().synchronize()

    def counts(self, b):
        """Record count per rank of the last completed gather of buffer b (on its root)."""
        import torch
        src = self.landing[b] if self.world > 1 else self.bufs[b]
        return src.view(self.world, self.stride)[:, :8].contiguous().view(torch.int64) \
            .flatten().cpu().tolist()

    def rows(self, b):
        """fs_row records of all ranks in rank order (numpy, on the gather's root): the 8- or 16-byte
        wire records expanded on the GPU, work indices local to each rank's batch.
        Returns (rows, per-rank counts)."""
        import torch
        cnts = self.counts(b)
        src = self.landing[b] if self.world > 1 else self.bufs[b]
        if self.rec_bytes != 32 and src.device.type != "cuda":
            src = src.cuda()                        # the expansion kernels need the records in HBM
        if self.rec_bytes != 32:
            from .engine import torch_ready
            torch_ready()                           # (torch's copy is complete before the library's stream reads it)
        parts = []
        for r in range(self.world):
            n = min(int(cnts[r]), self.cap)
            at = src.data_ptr() + r * self.stride + HDR
            if self.rec_bytes == 32:
                parts.append(src[r * self.stride + HDR:r * self.stride + HDR + n * 32]
                             .cpu().numpy().view(abi.ROW_DTYPE).copy())
                continue
            full = torch.empty(max(1, n) * 32, dtype=torch.uint8, device="cuda")
            if self.rec_bytes == 8:
                base = self.all_off.data_ptr() + r * (self.off_len + 1) * 8
                n_works = int(self.all_off[r * (self.off_len + 1) + self.off_len].item())
                self.index.unpack8_device(at, n, base, n_works, full.data_ptr())
            else:
                self.index.unpack_device(at, n, full.data_ptr())
            parts.append(full[:n * 32].cpu().numpy().view(abi.ROW_DTYPE).copy())
        return parts, cnts


SHARD_MAGIC = 0x46535348          # "FSSH"
SHARD_HDR = 64                    # eight int64: magic, rec_bytes, n_rows, n_works, word bytes, 0, 0, 0


def pack_shard(shard, device):
    """One rank's share of a batch as ONE self-describing byte tensor on `device`:
    [64-byte header | work offsets (n_works + 1 uint64) | records | fan words joined by
    '\n' (UTF-8)].  The record format travels in the header, so ranks need not agree on
    it (a shard with an out-of-vocabulary token carries fs_row records, its neighbour
    8-byte wire records, an empty one nothing).  The records are sliced out of the
    search's own buffer: on a GPU they go from the kernel to the collective without
    visiting the host."""
    import torch
    words = "\n".join(shard.words).encode("utf-8")
    off = np.ascontiguousarray(shard.work_off, dtype=np.uint64)
    hdr = np.zeros(SHARD_HDR // 8, dtype=np.int64)
    hdr[:5] = (SHARD_MAGIC, shard.rec_bytes, shard.n_rows, len(off) - 1, len(words))
    head = np.concatenate([hdr.view(np.uint8), off.view(np.uint8),
                           np.zeros((-off.nbytes) % 16, dtype=np.uint8)])   # records 16-byte aligned
    n = shard.n_rows * shard.rec_bytes
    parts = [torch.from_numpy(head).to(device), shard.buf[HDR:HDR + n].to(device)]
    if words:
        parts.append(torch.frombuffer(bytearray(words), dtype=torch.uint8).to(device))
    pad = (-sum(int(p.numel()) for p in parts)) % 16      # (the next rank's slot stays 16-byte aligned)
    if pad:
        parts.append(torch.zeros(pad, dtype=torch.uint8, device=device))
    return torch.cat(parts)


def unpack_shard(blob, index):
    """(fs_row records, fan words) of one rank's payload (rank 0).  Wire records are
    expanded by the library on the GPU (fs_rows_unpack8 / fs_rows_unpack)."""
    import torch
    hdr = blob[:SHARD_HDR].cpu().numpy().view(np.int64)
    if int(hdr[0]) != SHARD_MAGIC:
        raise RuntimeError("shard payload without its header (magic %#x)" % int(hdr[0]))
    rec, n, n_works, wbytes = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4])
    at_off = SHARD_HDR
    at_rec = at_off + ((8 * (n_works + 1) + 15) & ~15)
    at_words = at_rec + n * rec
    if n == 0:
        rows = np.zeros(0, dtype=abi.ROW_DTYPE)
    elif rec == 32:
        rows = blob[at_rec:at_words].cpu().numpy().view(abi.ROW_DTYPE).copy()
    elif rec in (8, 16):
        if index is None:
            raise RuntimeError("wire records need the script index to be expanded")
        dev = blob if blob.device.type == "cuda" else blob[:at_words].cuda()
        full = torch.empty(max(1, n) * 32, dtype=torch.uint8, device="cuda")
        from .engine import torch_ready
        if rec == 8:
            # (uint64 offsets: an 8-byte aligned copy of their own)
            offs = dev[at_off:at_off + 8 * (n_works + 1)].clone()
            torch_ready()                           # the copy and the clone are torch's, the expansion runs on the library's stream
            index.unpack8_device(dev.data_ptr() + at_rec, n, offs.data_ptr(), n_works, full.data_ptr())
        else:
            torch_ready()
            index.unpack_device(dev.data_ptr() + at_rec, n, full.data_ptr())
        rows = full[:n * 32].cpu().numpy().view(abi.ROW_DTYPE).copy()
    else:
        raise RuntimeError("shard payload with %d-byte records" % rec)
    words = bytes(blob[at_words:at_words + wbytes].cpu().numpy()).decode("utf-8").split("\n") if n else []
    if len(words) != len(rows):
        raise RuntimeError("gathered %d fan words for %d records" % (len(words), len(rows)))
    return rows, words


class Shard(object):
    """What one rank's search of its share of a batch leaves behind.
      buf        torch uint8 [32-byte header | records]: the count in the header's first
                 eight bytes, records of `rec_bytes` (32: fs_row, 16 / 8: wire records)
      work_off   token offsets of the shard's works (numpy uint64; for 8-byte records)
      words      fan word text per record, in record order
    """

    def __init__(self, buf, rec_bytes, n_rows, work_off, words):
        self.buf, self.rec_bytes, self.n_rows = buf, int(rec_bytes), int(n_rows)
        self.work_off, self.words = work_off, words


def shard_from_rows(rows, words, n_works):
    """A Shard of host fs_row records (searchers without a device path)."""
    import torch
    rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
    buf = torch.zeros(HDR + max(1, len(rows)) * 32, dtype=torch.uint8)
    buf[:8] = torch.from_numpy(np.array([len(rows)], dtype=np.uint64).view(np.uint8))
    if len(rows):
        buf[HDR:HDR + rows.nbytes] = torch.from_numpy(rows.view(np.uint8).reshape(-1).copy())
    return Shard(buf, 32, len(rows), np.zeros(n_works + 1, dtype=np.uint64), list(words))


class RankFailed(RuntimeError):
    """Another rank failed in its share of a batch; this rank stops too."""


def search_sharded(filenames, weights, searcher, group=None, root=0):
    """Search one batch of works across all ranks.

    `searcher.search_shard(sub_filenames) -> Shard` (AnnIndexSearch: records left in HBM)
    or, without it, `searcher.search_rows(sub_filenames) -> (rows, words)`.  Every rank
    packs its shard into one self-describing payload (pack_shard); the exchange is
    TWO collectives per batch: an 8-byte all_reduce(MAX) of {payload size, failure flag}
    -- RCCL has no gatherv, so the ranks must agree on the padded size, and the same
    word tells every rank when one of them failed, so that nobody waits in a gather for
    a rank that has raised -- and ONE padded gather of the payloads to rank `root`.
    Returns on that rank (fs_row records with work indices into `filenames`, fan words),
    elsewhere (None, None).  analyze() lets the root go round (batch i to rank i mod N): on
    xGMI every pair of GPUs has its own link, so the records of consecutive batches arrive
    over different links instead of all over the seven that end at rank 0, and the ranks
    take turns at joining and writing a batch's CSV."""
    import torch
    import torch.distributed as dist
    live = dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    bounds = split_contiguous(weights, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    sub = filenames[lo:hi]
    on_gpu = live and dist.get_backend(group) == "nccl"
    cdev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    payload, failure = None, None
    try:
        if hasattr(searcher, "search_shard"):
            shard = searcher.search_shard(sub)
        else:
            rows, words = searcher.search_rows(sub)
            shard = shard_from_rows(rows, words, len(sub))
        payload = pack_shard(shard, cdev)
    except Exception as e:                      # told to the other ranks below, then raised
        failure = e
    if world > 1:
        agree = torch.tensor([0 if payload is None else int(payload.numel()), 1 if failure else 0],
                             dtype=torch.int64, device=cdev)
        dist.all_reduce(agree, op=dist.ReduceOp.MAX, group=group)
        size, failed = (int(v) for v in agree.cpu().tolist())
    else:
        size, failed = (0 if payload is None else int(payload.numel())), (1 if failure else 0)
    if failure is not None:
        raise failure
    if failed:
        raise RankFailed("another rank failed in its share of this batch")
    if world == 1:
        blobs = [payload]
    else:
        send = torch.zeros(size, dtype=torch.uint8, device=cdev)
        send[:payload.numel()] = payload
        recv = [torch.empty_like(send) for _ in range(world)] if rank == root else None
        dist.gather(send, recv, dst=root, group=group)
        blobs = recv
    if rank != root:
        return None, None
    engine = getattr(searcher, "engine", None)
    all_rows, all_words = [], []
    for r, blob in enumerate(blobs):
        part, words = unpack_shard(blob, engine)
        part["work"] += np.uint32(bounds[r])
        all_rows.append(part)
        all_words += words
    rows = np.concatenate(all_rows) if all_rows else np.zeros(0, dtype=abi.ROW_DTYPE)
    return rows, all_words


def collect_batch_files(pattern, n_batches, mine, failure, t_start, group=None):
    """End of a sharded `search` run: every rank has written the batch CSVs of the batches
    it was the root of (`mine`, file names `pattern.format(i)`), rank 0 is about to
    concatenate all `n_batches` of them into the dated file (search.analyze).

    One all_gather of {failure, sizes of my files}: a rank whose writer failed (`failure`)
    raises it, every other rank raises RankFailed -- nobody waits in a barrier for a rank
    that has left.  Rank 0 then checks that every batch file is in ITS directory with the
    size its writer reports and not older than this run (ranks of one node share the
    working directory; ranks with directories of their own, or on several nodes, do not);
    what is missing or stale is sent over: a broadcast of the list, a gather of the bytes,
    written by rank 0 under the same names."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    info = {"failed": None if failure is None else repr(failure), "sizes": {}}
    if failure is None:
        try:
            info["sizes"] = {i: os.path.getsize(pattern.format(i)) for i in mine}
        except OSError as e:
            failure = e
            info["failed"] = repr(e)
    infos = [None] * world
    dist.all_gather_object(infos, info, group=group)
    if failure is not None:
        raise failure
    bad = [r for r, x in enumerate(infos) if x["failed"]]
    if bad:
        raise RankFailed("rank %d failed writing its batch files: %s" % (bad[0], infos[bad[0]]["failed"]))
    need = [None]
    if rank == 0:
        sizes = {}
        for x in infos:
            sizes.update(x["sizes"])
        lost = [i for i in range(n_batches) if i not in sizes]
        if lost:
            need = [RuntimeError("no rank wrote batch %d" % lost[0])]
        else:
            need = [[]]
            for i in range(n_batches):
                try:
                    st = os.stat(pattern.format(i))
                    # a batch this rank wrote itself in this run is the file, whatever the file
                    # system's clock says (nobody could send a better one): the size decides
                    ok = st.st_size == sizes[i] and (i in info["sizes"] or st.st_mtime >= t_start - 2.0)
                except OSError:
                    ok = False
                if not ok:
                    need[0].append(i)
    dist.broadcast_object_list(need, src=0, group=group)
    need = need[0]
    if isinstance(need, Exception):
        raise need
    if not need:
        return
    parts = {}
    for i in need:
        if i in info["sizes"] and rank != 0:
            with open(pattern.format(i), "rb") as fh:
                parts[i] = fh.read()
    got = [None] * world if rank == 0 else None
    dist.gather_object(parts, got, dst=0, group=group)
    if rank == 0:
        for x in got:
            for i, data in x.items():
                with open(pattern.format(i), "wb") as fh:
                    fh.write(data)
        still = [i for i in need if not any(i in x for x in got)]
        if still:
            raise RuntimeError("batch file %s is stale or missing and no other rank holds it"
                               % pattern.format(still[0]))
