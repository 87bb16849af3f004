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
