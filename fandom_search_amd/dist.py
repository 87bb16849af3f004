"""Multi-GPU execution of the search: one process per GPU, works sharded,
match rows gathered to rank 0.

The reference's only parallelism is a 4-process multiprocessing.Pool over the
works of a batch whose results are concatenated in input order
(/root/reference/search.py:381-386).  Here the works of a batch are split into
contiguous ranges, rank r searches range r on its own GPU against a replicated
script index (built from the same inputs on every rank: no broadcast), and the
variable-length row buffers are gathered to rank 0 -- an all_gather of one
count per rank, then a padded gather of `count_r x 32 B` (RCCL has no gatherv).
Rank 0 concatenates in rank order == work order, so the output bytes do not
depend on the number of GPUs.

Backend: "nccl" (= RCCL over xGMI) when a GPU is present, "gloo" otherwise
(CPU tests).
"""

import os

import numpy as np

from . import abi


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when
    WORLD_SIZE > 1.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
    return rank, local_rank, world


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


def gather_rows(rows, group=None, dst=0):
    """Gather fs_row arrays (numpy, abi.ROW_DTYPE) of all ranks to `dst`, in
    rank order.  Returns the concatenated array on dst, None elsewhere."""
    import torch
    import torch.distributed as dist

    rows = np.ascontiguousarray(rows, dtype=abi.ROW_DTYPE)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")

    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([len(rows)], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts = counts.cpu().tolist()
    pad = max(counts)
    if pad == 0:
        return np.zeros(0, dtype=abi.ROW_DTYPE) if rank == dst else None
    send = torch.zeros(pad * abi.ROW_DTYPE.itemsize, dtype=torch.uint8)
    send[:rows.nbytes] = torch.from_numpy(rows.view(np.uint8).reshape(-1))
    send = send.to(dev)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [recv[r][:counts[r] * abi.ROW_DTYPE.itemsize].cpu().numpy().view(abi.ROW_DTYPE)
             for r in range(world)]
    return np.concatenate(parts)


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

    def __init__(self, index, cap, rec_bytes, n_buffers=1, group=None, rehearsal=False):
        import torch
        import torch.distributed as dist
        self.index, self.rec_bytes, self.group, self.rehearsal = index, int(rec_bytes), group, rehearsal
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.cdev = "cpu" if rehearsal else "cuda"
        if self.world > 1:                     # one capacity for all ranks (padded gather)
            t = torch.tensor([int(cap)], dtype=torch.int64, device=self.cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            cap = int(t.item())
        self.cap = int(cap)
        self.stride = HDR + self.cap * self.rec_bytes
        self.bufs = [torch.zeros(self.stride, dtype=torch.uint8, device="cuda")
                     for _ in range(n_buffers)]
        self.landing = None
        if self.world > 1 and self.rank == 0:
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
            self.all_off = mine.cuda()
            return
        mine = mine.to(self.cdev)
        out = torch.zeros(self.world * (self.off_len + 1), dtype=torch.int64, device=self.cdev)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        self.all_off = out.cuda() if self.rank == 0 else None

    def start(self, b):
        """Queue the gather of buffer b (after the search that fills it has ended)."""
        import torch.distributed as dist
        if self.world == 1:
            return
        send = self.bufs[b].cpu() if self.rehearsal else self.bufs[b]
        recv = list(self.landing[b].chunk(self.world)) if self.rank == 0 else None
        self.pending[b] = dist.gather(send, recv, dst=0, group=self.group, async_op=True)

    def wait(self, b):
        """Buffer b may be written again / read on rank 0 after this.  Work.wait() on
        RCCL only orders torch's current stream, and the library writes the buffers
        from its own streams, so the host also waits for that stream."""
        import torch
        if self.pending[b] is None:
            return
        self.pending[b].wait()
        self.pending[b] = None
        if not self.rehearsal:
            torch.cuda.current_stream().synchronize()

    def counts(self, b):
        """Record count per rank of the last completed gather of buffer b (rank 0)."""
        import torch
        src = self.landing[b] if self.world > 1 else self.bufs[b]
        return src.view(self.world, self.stride)[:, :8].contiguous().view(torch.int64) \
            .flatten().cpu().tolist()

    def rows(self, b):
        """fs_row records of all ranks in rank order (numpy, rank 0): the 8- or 16-byte
        wire records expanded on the GPU, work indices local to each rank's batch.
        Returns (rows, per-rank counts)."""
        import torch
        cnts = self.counts(b)
        src = self.landing[b] if self.world > 1 else self.bufs[b]
        if self.rehearsal and self.world > 1:
            src = src.cuda()
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


def gather_strings(items, group=None, dst=0):
    """Gather a list of Python strings per rank to `dst` (rank order)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(items)
    world = dist.get_world_size(group)
    out = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(list(items), out, dst=dst, group=group)
    if out is None:
        return None
    return [s for part in out for s in part]


def search_sharded(filenames, weights, search_rows, group=None):
    """Search one batch of works across all ranks.

    search_rows(sub_filenames) -> (rows with work indices local to the sub
    list, fan word text per row).  Returns on rank 0 (rows with work indices
    into `filenames`, fan words), elsewhere (None, None)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bounds = split_contiguous(weights, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    rows, words = search_rows(filenames[lo:hi])
    rows = np.array(rows, dtype=abi.ROW_DTYPE, copy=True)
    rows["work"] += np.uint32(lo)
    all_rows = gather_rows(rows, group=group)
    all_words = gather_strings(words, group=group)
    if rank != 0:
        return None, None
    return all_rows, all_words
