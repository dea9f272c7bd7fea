"""Multi-GPU layer of `identify`: reads shard by rank, the index is replicated, and the only exchange
is the sum of the per-rank profile tables at the end of a file -- the cross-thread reduce of
source/modes/Compare.hpp:3445-3454 done across GPUs with one RCCL all-reduce over xGMI.

The tables travel as integer limbs (kasa_profile_export_limbs: {unique, total, 4 x 32-bit limbs of the
64.64 fixed-point countAll}), so the sum is exact and independent of rank order; carries are
normalised after the reduce.  Backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors
(tests/test_dist_cpu.py, world_size 2).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_reads: int, rank: int, world: int):
    """Contiguous read range of a rank, so per-read outputs concatenate in input order."""
    base, rem = divmod(n_reads, world)
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


def allreduce_limbs(limbs: np.ndarray, device=None) -> np.ndarray:
    """Sum u64 limb tables over all ranks of the default process group (no-op without one)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return limbs
    t = torch.from_numpy(limbs.astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def all_to_all_arrays(send):
    """send[j] = 1-D numpy array for rank j -> list of the arrays every rank sent to this one (same dtype).
    RCCL (`nccl`) moves them as device tensors with one all_to_all; gloo has no all-to-all, there the same exchange
    runs as point-to-point transfers (CPU tests)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dtype = send[0].dtype
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    sizes = torch.tensor([int(a.shape[0]) for a in send], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    incoming = [int(all_sizes[s][rank]) for s in range(world)]
    tx = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).to(dev) for a in send]
    rx = [torch.zeros(n * dtype.itemsize, dtype=torch.uint8, device=dev) for n in incoming]
    if nccl:
        dist.all_to_all(rx, tx)
    else:
        rx[rank].copy_(tx[rank])
        reqs = []
        for peer in range(world):
            if peer == rank:
                continue
            reqs.append(dist.isend(tx[peer], peer))
            reqs.append(dist.irecv(rx[peer], peer))
        for q in reqs:
            q.wait()
    return [t.cpu().numpy().view(dtype) for t in rx]


def partitioned_batch(owner_ctx, worker, cuts, K: int, batch, want_per_read: bool = True, unique: bool = False):
    """One batch against an index that is range-partitioned over the ranks (rank j holds partition j; see
    kasa_amd/partition.py): two exchanges of query slices and two of event records.  Returns `owner_ctx`, scored."""
    import torch.distributed as dist
    from . import partition
    world = dist.get_world_size()
    ctx = owner_ctx
    ctx.upload(batch.bases, batch.offsets, batch.seg_read, batch.n if batch.seg_read is not None else None)
    ctx.encode()
    ctx.sort_and_range(unique)
    km, rd = ctx.queries()
    starts = partition.slice_starts(km, cuts, K)
    km_in = all_to_all_arrays([km[starts[j]:starts[j + 1]] for j in range(world)])
    rd_in = all_to_all_arrays([rd[starts[j]:starts[j + 1]] for j in range(world)])
    n_in = all_to_all_arrays([np.asarray([ctx.n_reads], dtype=np.int64) for _ in range(world)])
    rec_out, pool_out = [], []
    for s in range(world):                                        # the slices of every rank, against my partition
        rec, pool = worker.group_slice(km_in[s], rd_in[s], int(n_in[s][0]))
        rec_out.append(rec.reshape(-1))
        pool_out.append(pool)
    rec_back = all_to_all_arrays(rec_out)
    pool_back = all_to_all_arrays(pool_out)
    parts = [(rec_back[j].reshape(-1, ctx.rec_words), pool_back[j]) for j in range(world)]
    rec, pool = partition.assemble_records(parts, starts)
    ctx.records_import(rec, pool)
    ctx.score(want_per_read)
    return ctx


def limbs_to_tables(limbs: np.ndarray, n_k: int, n_taxa: int):
    """limbs[nK*nTaxa, 6] -> (countAll f64, countUnique u64, countTotal u64), each [nK, nTaxa]."""
    limbs = limbs.reshape(n_k * n_taxa, 6)
    unique = limbs[:, 0].copy()
    total = limbs[:, 1].copy()
    all_ = np.zeros(limbs.shape[0], dtype=np.float64)
    for i in range(limbs.shape[0]):
        v = int(limbs[i, 2]) + (int(limbs[i, 3]) << 32) + (int(limbs[i, 4]) << 64) + (int(limbs[i, 5]) << 96)
        all_[i] = float(v >> 64) + float(v & ((1 << 64) - 1)) * 2.0 ** -64
    sh = (n_k, n_taxa)
    return all_.reshape(sh), unique.reshape(sh), total.reshape(sh)
