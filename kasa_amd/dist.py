"""Multi-GPU layer of `identify`: reads shard by rank, the index is replicated, and the only exchange
is the sum of the per-rank profile tables at the end of a file -- the cross-thread reduce of
source/modes/Compare.hpp:3445-3454 done across GPUs with one RCCL all-reduce over xGMI.

The tables travel as integer limbs (kasa_profile_export_limbs: {unique, total, 4 x 32-bit limbs of the
64.64 fixed-point countAll}), so the sum is exact and independent of rank order; carries are
normalised after the reduce.  Backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors
(tests/test_dist_cpu.py, world_size 2).
"""
from __future__ import annotations

import os

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


class _NcclUniqueId(__import__("ctypes").Structure):
    _fields_ = [("internal", __import__("ctypes").c_char * 128)]


_rccl = None


def rccl_lib():
    """The RCCL library of this process (the one libkasa_hip.so resolved its ncclAllReduce to)."""
    global _rccl
    import ctypes as C
    import os
    if _rccl is None:
        err = None
        cands = ["librccl.so.1", "librccl.so"]
        try:
            import torch
            cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        except Exception:
            pass
        cands.append("/opt/rocm/lib/librccl.so")
        for name in cands:
            try:
                _rccl = C.CDLL(name)
                break
            except OSError as e:
                err = e
        if _rccl is None:
            raise RuntimeError(f"librccl not found: {err}")
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        _rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        _rccl.ncclGetErrorString.restype = C.c_char_p
    return _rccl


def rccl_communicator(rank: int, world: int):
    """An RCCL communicator of its own for the C ABI (kasa_profile_allreduce(ctx, ncclComm_t)): rank 0 draws the
    ncclUniqueId, torch.distributed's default group (any backend) carries its 128 bytes to the others, everybody calls
    ncclCommInitRank on its current device.  -> (comm as int, number of ranks RCCL reports)."""
    import ctypes as C
    import torch.distributed as dist
    L = rccl_lib()
    uid = _NcclUniqueId()
    if rank == 0:
        rc = L.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError("ncclGetUniqueId: " + L.ncclGetErrorString(rc).decode())
    box = [bytes(uid) if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    C.memmove(C.byref(uid), box[0], 128)
    comm = C.c_void_p()
    # ncclCommInitRank is a collective: a rank that never arrives leaves the others waiting for ever.  It runs on a helper
    # thread with a deadline (KASA_RCCL_INIT_TIMEOUT seconds, default 180): a caller that gets the exception falls back to
    # another reduce or ends the run -- it does not hang.
    import threading
    res = {}

    def init():
        res["rc"] = L.ncclCommInitRank(C.byref(comm), world, uid, rank)
    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(float(os.environ.get("KASA_RCCL_INIT_TIMEOUT", "180")))
    if th.is_alive():
        raise TimeoutError(f"ncclCommInitRank did not return within the deadline on rank {rank} of {world}")
    rc = res.get("rc", -1)
    if rc != 0:
        raise RuntimeError("ncclCommInitRank: " + L.ncclGetErrorString(rc).decode())
    n = C.c_int(0)
    L.ncclCommCount(comm, C.byref(n))
    return int(comm.value), int(n.value)


def rccl_destroy(comm: int):
    import ctypes as C
    if comm:
        rccl_lib().ncclCommDestroy(C.c_void_p(comm))


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


_hip = None


def _device_copy(dst: int, src: int, nbytes: int):
    """hipMemcpy between two device pointers (the library's buffers <-> torch tensors)."""
    global _hip
    import ctypes as C
    if nbytes == 0:
        return
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    if not dst or not src:
        raise RuntimeError("device copy with a NULL pointer")
    rc = _hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 4)   # hipMemcpyDefault
    if rc == 0:
        rc = _hip.hipDeviceSynchronize()                            # a device-to-device hipMemcpy may return before the copy has landed
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed ({rc})")


def _all_to_all_device(send, send_counts, unit: int, recv=None, recv_counts=None):
    """One all_to_all of byte tensors on the device: send = uint8 tensor holding the slices for rank 0, 1, ... back to
    back, send_counts[j] = elements of `unit` bytes for rank j.  Returns (recv tensor, counts received from every rank).
    recv / recv_counts: a tensor to receive into when the counts are known beforehand (else they are exchanged first)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = send.device
    if recv_counts is None:
        mine = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        theirs = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(theirs, mine)
        recv_counts = [int(x) for x in theirs.cpu()]
    if recv is None:
        recv = torch.empty(sum(recv_counts) * unit, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv, send, [n * unit for n in recv_counts], [n * unit for n in send_counts])
    # The collective only orders torch's current stream behind RCCL's; the library reads `recv` through raw pointers on its
    # own non-blocking stream.  Wait for the data before handing the pointers on (and before `send` can be released).
    torch.cuda.current_stream(dev).synchronize()
    return recv, list(recv_counts)


class _DevView:
    """A device pointer as an object torch can wrap without copying (__cuda_array_interface__)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


def _device_view(ptr: int, nbytes: int, dev):
    """uint8 tensor over `nbytes` of device memory at `ptr` (no copy; the owner keeps it alive), or None if torch refuses."""
    import torch
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=dev)
    try:
        t = torch.as_tensor(_DevView(ptr, nbytes), device=dev)
        return t if (t.data_ptr() == ptr and t.numel() == nbytes) else None
    except Exception:
        return None


def partitioned_batch_device(owner_ctx, worker, cuts, batch, want_per_read: bool = True, unique: bool = False, stats=None):
    """partitioned_batch with every slice and every record staying in HBM: the sorted k-mers travel straight out of the
    library's buffer (kasa_batch_queries_device) in ONE RCCL all_to_all (8 or 16 bytes per query; the read ids stay at home
    -- grouping does not need them), the event records come back in a second one straight into the context's inbox
    (kasa_batch_records_inbox: shifted and filed from there in place), the taxon lists in a third.  Nothing of the
    exchange touches host memory and no record is copied more often than the collective itself does.
    stats (dict): bytes this rank sent and received."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    ctx = owner_ctx
    ctx.upload(batch.bases, batch.offsets, batch.seg_read, batch.n if batch.seg_read is not None else None)
    ctx.encode()
    ctx.sort_and_range(unique)
    ptr, n, kb = ctx.queries_device()
    starts = ctx.slice_starts(cuts)
    send = _device_view(ptr, n * kb, dev)
    if send is None:                                              # (a torch without the array interface: one copy)
        send = torch.empty(n * kb, dtype=torch.uint8, device=dev)
        _device_copy(send.data_ptr(), ptr, n * kb)
    out_counts = [int(starts[j + 1] - starts[j]) for j in range(world)]
    km_in, n_in = _all_to_all_device(send, out_counts, kb)
    del send
    rw = ctx.rec_words * 4
    total_in = int(sum(n_in))
    rec_out = torch.empty(total_in * rw, dtype=torch.uint8, device=dev)
    pools, pool_counts, at = [], [], 0
    for s in range(world):                                        # the slices of every rank, against my partition
        # grouped straight into the send tensor (kasa_batch_group_to: no copy of the records); the slice's profile stays on
        # this rank (the reduce sums the ranks).  records_device() waits for the worker's stream -- not for the device
        rp, nrw, pp, npw = worker.group_slice_device(km_in.data_ptr() + at * kb, n_in[s], sink=ctx, records_out=rec_out.data_ptr() + at * rw)
        if n_in[s] == 0 and nrw:
            _device_copy(rec_out.data_ptr() + at * rw, rp, nrw * 4)
        p = torch.empty(npw * 4, dtype=torch.uint8, device=dev)   # the pool is the worker's own buffer, reused by the next slice: a small copy
        src = _device_view(pp, npw * 4, dev)
        if src is not None:
            p.copy_(src)
            torch.cuda.current_stream(dev).synchronize()          # (before the worker overwrites its pool)
        else:
            _device_copy(p.data_ptr(), pp, npw * 4)
        pools.append(p); pool_counts.append(npw)
        at += n_in[s]
    del km_in
    # records: PACKED for the wire (kasa_batch_records_pack: classes + the words a record really uses -- matched queries only,
    # 16 / 24 / 28 of a narrow record's 32 bytes by its number of segments), unpacked on arrival into the place
    # kasa_batch_records_import_device files them from (the context's own staging buffer), shifted there in place
    inbox = ctx.records_inbox(n * ctx.rec_words)
    compact = os.environ.get("KASA_WIRE_WHOLE_RECORDS") != "1"
    if compact:
        sizes, at = [], 0
        for s in range(world):
            sizes.append(worker.ctx.records_pack_size(rec_out.data_ptr() + at * rw, n_in[s]))
            at += n_in[s]
        packed = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        at, pa_ = 0, 0
        for s in range(world):
            worker.ctx.records_pack_size(rec_out.data_ptr() + at * rw, n_in[s])  # (the block offsets of THIS slice once more: one pass over its records)
            worker.ctx.records_pack(rec_out.data_ptr() + at * rw, n_in[s], packed.data_ptr() + pa_, sizes[s])
            at += n_in[s]; pa_ += sizes[s]
        del rec_out
        packed_back, bytes_in = _all_to_all_device(packed, sizes, 1)
        del packed
        pa_, ra_ = 0, 0
        for j in range(world):
            ctx.records_unpack(packed_back.data_ptr() + pa_, bytes_in[j], out_counts[j], inbox + ra_ * rw)
            pa_ += bytes_in[j]; ra_ += out_counts[j]
        rec_n = list(out_counts)
        rec_base = inbox
        wire_sent, wire_recv = sizes, bytes_in
        del packed_back
    else:
        rec_recv = _device_view(inbox, n * rw, dev)
        rec_back, rec_n = _all_to_all_device(rec_out, n_in, rw, recv=rec_recv, recv_counts=out_counts)
        del rec_out
        rec_base = rec_back.data_ptr()
        wire_sent, wire_recv = [c * rw for c in n_in], [c * rw for c in rec_n]
    pool_back, pool_n = _all_to_all_device(torch.cat(pools) if pools else torch.empty(0, dtype=torch.uint8, device=dev), pool_counts, 4)
    del pools
    parts, ra, pa = [], 0, 0
    for j in range(world):
        parts.append((rec_base + ra * rw, rec_n[j] * ctx.rec_words, pool_back.data_ptr() + pa * 4, pool_n[j]))
        ra += rec_n[j]; pa += pool_n[j]
    ctx.records_import_device(parts)
    if stats is not None:
        me = dist.get_rank()
        stats.update({"queries_sent": sum(c for j, c in enumerate(out_counts) if j != me) * kb,
                      "queries_received": sum(c for j, c in enumerate(n_in) if j != me) * kb,
                      "records_sent": sum(c for j, c in enumerate(wire_sent) if j != me),
                      "records_received": sum(c for j, c in enumerate(wire_recv) if j != me),
                      "records_whole": sum(c for j, c in enumerate(rec_n) if j != me) * rw,
                      "records_on_the_wire": "packed (classes + used words)" if compact else "whole",
                      "pool_sent": sum(c for j, c in enumerate(pool_counts) if j != me) * 4,
                      "pool_received": sum(c for j, c in enumerate(pool_n) if j != me) * 4})
    del pool_back
    ctx.score(want_per_read)
    return ctx


def partitioned_batch(owner_ctx, worker, cuts, K: int, batch, want_per_read: bool = True, unique: bool = False, stats=None):
    """One batch against an index that is range-partitioned over the ranks (rank j holds partition j; see
    kasa_amd/partition.py): two exchanges of query slices and two of event records.  Returns `owner_ctx`, scored.
    With RCCL (`nccl`) the exchange is device-resident (partitioned_batch_device); the host-staged form below serves
    gloo (CPU tensors, tests)."""
    import torch.distributed as dist
    from . import partition
    if dist.get_backend() == "nccl":
        return partitioned_batch_device(owner_ctx, worker, cuts, batch, want_per_read, unique, stats)
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
        rec, pool = worker.group_slice(km_in[s], rd_in[s], int(n_in[s][0]), sink=ctx)   # (the slice's profile stays on this rank: the reduce sums the ranks)
        rec_out.append(rec.reshape(-1))
        pool_out.append(pool)
    # the records travel packed (partition.pack_records: the numpy statement of kasa_batch_records_pack's wire format)
    rw_words = ctx.rec_words
    wire_back = all_to_all_arrays([partition.pack_records(r.reshape(-1, rw_words), rw_words) for r in rec_out])
    out_n = [int(starts[j + 1] - starts[j]) for j in range(world)]
    rec_back = [partition.unpack_records(wire_back[j], out_n[j], rw_words).reshape(-1) for j in range(world)]
    pool_back = all_to_all_arrays(pool_out)
    if stats is not None:
        me = dist.get_rank()
        stats.update({"queries_sent": sum(int(km[starts[j]:starts[j + 1]].nbytes) for j in range(world) if j != me),
                      "records_received": sum(int(wire_back[j].nbytes) for j in range(world) if j != me),
                      "records_whole": sum(int(rec_back[j].nbytes) for j in range(world) if j != me),
                      "pool_received": sum(int(pool_back[j].nbytes) for j in range(world) if j != me)})
    parts = [(rec_back[j].reshape(-1, ctx.rec_words), pool_back[j]) for j in range(world)]
    rec, pool = partition.assemble_records(parts, starts)
    ctx.records_import(rec, pool)
    ctx.score(want_per_read)
    return ctx


def partitioned_batches(owner_ctx, worker, cuts, K: int, batches, want_per_read: bool = True, unique: bool = False):
    """partitioned_batch for a rank's whole list of batches.  The exchange is a collective: ranks may hold different numbers
    of batches (read shards of different sizes, other `-m` cuts), so before every round the ranks agree (one all-reduce of
    a flag) whether anybody still has a batch, and a rank that has run out takes part with an empty one -- its partition
    is still needed by the others.  Yields `owner_ctx` after each of this rank's own batches."""
    import torch
    import torch.distributed as dist
    from .reads import ReadBatch
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    it = iter(batches)
    empty = ReadBatch(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64), [], np.zeros(0, dtype=np.uint32))
    while True:
        batch = next(it, None)
        flag = torch.tensor([1 if batch is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) == 0:
            return
        ctx = partitioned_batch(owner_ctx, worker, cuts, K, batch if batch is not None else empty, want_per_read, unique)
        if batch is not None:
            yield ctx


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
