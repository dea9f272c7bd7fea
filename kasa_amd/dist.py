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
