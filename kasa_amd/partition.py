"""Range-partitioned index (BASELINE.json config C5: an index larger than one GPU's HBM).

The sorted index is cut at 30-bit prefix boundaries -- a prefix range of the reference's `_trie`
(source/modes/Trie.hpp:494-520) never straddles a cut -- and every partition is an index of its own
(`capi.DeviceIndex`).  One batch then runs as

    owner   : upload + encode + sort its reads                        (kasa_batch_upload/encode/sort_and_range)
              cut the sorted queries at the same prefixes: slice j is contiguous
    worker j: group slice j against partition j                       (kasa_batch_set_queries/sort_and_range/group,
                                                                        kasa_batch_records_fetch)
    owner   : concatenate the event records in partition order (= global sorted order), shift positions
              by the slice start and segment-list offsets by the pool offset, file them by read, score
                                                                        (kasa_batch_records_import/score)

A slice is a batch of its own for the worker: its first query opens a new prefix range, so nothing of
the grouping reaches across a cut, and the result equals the run against the unpartitioned index bit for
bit (tests/test_gpu_partition.py).  `LocalExchange` runs all partitions in one process (one GPU, tests);
`kasa_amd/dist.py:partitioned_batch` moves the slices with torch.distributed instead (one partition per
rank).
"""
from __future__ import annotations

import numpy as np

from . import capi, formats

def split_index(ix: formats.Index, n_parts: int):
    """-> (list of formats.Index, cuts u64[n_parts]): partition j holds the entries whose 30-bit prefix p
    satisfies cuts[j] <= p < cuts[j+1]; about equal record counts, cut only between `_trie` entries."""
    ends = np.cumsum(ix.trie_count.astype(np.int64))
    parts, cuts = [], []
    lo_t = 0
    for j in range(n_parts):
        target = ix.n * (j + 1) // n_parts
        hi_t = int(np.searchsorted(ends, target, side="left")) + 1 if j + 1 < n_parts else ends.shape[0]
        hi_t = max(min(hi_t, ends.shape[0]), lo_t)
        a = int(ends[lo_t - 1]) if lo_t else 0
        b = int(ends[hi_t - 1]) if hi_t else 0
        cuts.append(int(ix.trie_prefix[lo_t]) if lo_t < ends.shape[0] and j > 0 else (0 if j == 0 else 1 << 30))
        parts.append(formats.Index(ix.kmer[a:b], ix.taxid[a:b], ix.tax[a:b], ix.trie_prefix[lo_t:hi_t],
                                   ix.trie_count[lo_t:hi_t], ix.content, ix.freq))
        lo_t = hi_t
    return parts, np.asarray(cuts, dtype=np.uint64)


def slice_starts(km_sorted: np.ndarray, cuts: np.ndarray, K: int) -> np.ndarray:
    """Start of every partition's slice in the sorted queries (plus the end): i64[n_parts + 1]."""
    pre = formats.key_shr(km_sorted, 5 * (K - formats.TRIE_LETTERS))
    s = np.searchsorted(pre, cuts.astype(np.uint64), side="left").astype(np.int64)
    s[0] = 0
    return np.concatenate((s, [km_sorted.shape[0]]))


def assemble_records(parts, starts):
    """parts[j] = (rec u32[n_j, W], pool u32[m_j]) of slice j -> one (rec, pool) for the whole batch, sorted order.
    Record layout: include/kasa_hip.h (kasa_batch_group)."""
    recs, pools, base = [], [np.zeros(1, dtype=np.uint32)], 1
    for j, (rec, pool) in enumerate(parts):
        rec = rec.copy()
        w = rec.shape[1]
        inl, last = (4, 7) if w == 8 else (8, 15)                  # segments a record holds itself; word of the pool offset
        nseg = (rec[:, 3] & np.uint32(255)) if w == 8 else rec[:, 3]
        matched = (rec[:, 2] & np.uint32(31)) != 0
        rec[:, 0] += np.uint32(starts[j])                           # sorted position: slice-local -> batch
        rec[matched, 1] += np.uint32(starts[j])                     # last flush position
        lists = matched & (nseg > np.uint32(inl))                   # further segments live in the slice's pool (word 0 unused)
        rec[lists, last] += np.uint32(base - 1)
        recs.append(rec)
        pools.append(pool[1:])
        base += pool.shape[0] - 1
    w = parts[0][0].shape[1] if parts else 8
    rec_all = np.concatenate(recs) if recs else np.zeros((0, w), dtype=np.uint32)
    return rec_all, np.concatenate(pools)


class Worker:
    """Partition owner: groups slices of foreign sorted queries against its partition."""

    def __init__(self, dix: capi.DeviceIndex, k_high: int, k_low: int, frames: int):
        self.ctx = capi.Context(dix, k_high, k_low, frames)

    def group_slice(self, km: np.ndarray, rd: np.ndarray, n_reads: int, sink: capi.Context = None):
        """sink: the context that collects this rank's profile -- the profile of a slice is made where the slice is grouped
        (kasa_batch_group), i.e. here; the sum over the ranks' sinks is the file's profile."""
        self.ctx.set_queries(km, rd, n_reads)
        self.ctx.sort_and_range()                                  # already sorted; the sort is stable
        self.ctx.group()
        if sink is not None:
            sink.profile_absorb(self.ctx)
        return self.ctx.records()

    def group_slice_device(self, ptr: int, n: int, sink: capi.Context = None, records_out: int = 0):
        """The same for a slice that is already in device memory (`ptr`: its sorted k-mers); returns device pointers
        (records pointer, record words, pool pointer, pool words), valid until the worker's next slice.
        records_out: device memory the records are written to directly (the send buffer of the exchange)."""
        self.ctx.set_sorted_device(ptr, n)
        if records_out and n:
            self.ctx.group_to(records_out)
        else:
            self.ctx.group()
        if sink is not None:
            sink.profile_absorb(self.ctx)
        return self.ctx.records_device()

    def close(self):
        self.ctx.close()


class LocalExchange:
    """All partitions in this process (one GPU): the reference implementation of the exchange, used by the tests."""

    def __init__(self, parts, cuts, k_high=12, k_low=7, frames=3, device: int = 0, device_resident: bool = False, K: int = None, packed: bool = True):
        """parts: formats.Index objects, or capi.DeviceIndex objects that are on the device already (then K = letters per
        index k-mer must be given).  This is also how ONE device holds an index of more than 2^32 records (positions in an
        index are 32-bit, kasa_index_create refuses more): as several range partitions, every one an index of its own."""
        self.cuts = cuts
        self.device_resident = device_resident      # slices and records never leave HBM (kasa_batch_*_device)
        self.packed = packed                        # records cross between worker and owner in the wire format (matched queries only, used words only)
        self.wire_bytes = self.whole_bytes = 0      # of the last batch: bytes of records on the wire, bytes whole records would have taken
        given = len(parts) > 0 and isinstance(parts[0], capi.DeviceIndex)
        self.K = K if given else parts[0].K
        if self.K is None:
            raise ValueError("LocalExchange over device indices: K (12 or 25) must be given")
        self.dix = list(parts) if given else [capi.DeviceIndex(p, device) for p in parts]
        self.workers = [Worker(d, k_high, k_low, frames) for d in self.dix]
        self.owner = capi.Context(self.dix[0], k_high, k_low, frames)   # the owner needs an index only for its key width

    def run_batch(self, batch, want_per_read=True, unique=False):
        ctx = self.owner
        ctx.upload(batch.bases, batch.offsets, batch.seg_read, batch.n if batch.seg_read is not None else None)
        ctx.encode()
        ctx.sort_and_range(unique)
        rw = ctx.rec_words
        self.wire_bytes = self.whole_bytes = 0
        if self.device_resident:
            ptr, n, kb = ctx.queries_device()
            starts = ctx.slice_starts(self.cuts)
            if not self.packed:
                parts = [w.group_slice_device(ptr + int(starts[j]) * kb, int(starts[j + 1] - starts[j]), sink=ctx) for j, w in enumerate(self.workers)]
            else:
                # every slice's records go over the "wire" packed (kasa_batch_records_pack on the worker, _unpack on the owner,
                # into the inbox the whole records would have been received at)
                inbox, parts, at = ctx.records_inbox(n * rw), [], 0
                for j, w in enumerate(self.workers):
                    nq = int(starts[j + 1] - starts[j])
                    rp, nrw, pp, npw = w.group_slice_device(ptr + int(starts[j]) * kb, nq, sink=ctx)
                    nb = w.ctx.records_pack_size(rp, nq)
                    buf = capi.DeviceBuffer(nb, ctx.dix.device)
                    w.ctx.records_pack(rp, nq, buf.ptr, nb)
                    ctx.records_unpack(buf.ptr, nb, nq, inbox + at * rw * 4)
                    buf.close()
                    parts.append((inbox + at * rw * 4, nq * rw, pp, npw))
                    at += nq
                    self.wire_bytes += nb; self.whole_bytes += nq * rw * 4
            ctx.records_import_device(parts)
            ctx.score(want_per_read)
            return ctx
        km, rd = ctx.queries()
        starts = slice_starts(km, self.cuts, self.K)
        parts = [w.group_slice(km[starts[j]:starts[j + 1]], rd[starts[j]:starts[j + 1]], ctx.n_reads, sink=ctx)
                 for j, w in enumerate(self.workers)]
        if self.packed:                                             # (the numpy statement of the same wire format)
            packed = []
            for j, (rec_j, pool_j) in enumerate(parts):
                nq = int(starts[j + 1] - starts[j])
                wire = pack_records(np.asarray(rec_j).reshape(-1, rw), rw)
                packed.append((unpack_records(wire, nq, rw), pool_j))
                self.wire_bytes += int(wire.nbytes); self.whole_bytes += nq * rw * 4
            parts = packed
        rec, pool = assemble_records(parts, starts)
        ctx.records_import(rec, pool)
        ctx.score(want_per_read)
        return ctx

    def close(self):
        self.owner.close()
        for w in self.workers:
            w.close()
        for d in self.dix:
            d.close()


# ---- the wire format of exported records (kasa_batch_records_pack / _unpack in the C ABI; this is its statement in numpy, used
# by the host-staged exchange and as the cross-check of the device kernels): one byte of classes per four records (2 bits
# each: 0 = unmatched; 1, 2, 3 = words [1 .. n] of the record, n = 4 / 6 / 7 for 8-word records (at most one / at most three /
# more segments), 9 / 11 / 15 for 16-word ones (at most two / at most four / more)), padded to 16 bytes, then the words.
_WIRE_WORDS = {8: np.array([0, 4, 6, 7], dtype=np.int64), 16: np.array([0, 9, 11, 15], dtype=np.int64)}


def _wire_classes(rec: np.ndarray, rw: int) -> np.ndarray:
    d = rec[:, 2] & 31
    n = (rec[:, 3] & 255) if rw == 8 else rec[:, 3]
    lo, hi = (1, 3) if rw == 8 else (2, 4)
    cls = np.where(n <= lo, 1, np.where(n <= hi, 2, 3)).astype(np.uint8)
    cls[d == 0] = 0
    return cls


def pack_records(rec: np.ndarray, rw: int) -> np.ndarray:
    """rec[n, rw] (uint32, slice order) -> the bytes on the wire (uint8)."""
    rec = np.ascontiguousarray(rec, dtype=np.uint32).reshape(-1, rw)
    n = rec.shape[0]
    cls = _wire_classes(rec, rw)
    pad = np.zeros((-n) % 4, dtype=np.uint8)
    c4 = np.concatenate((cls, pad)).reshape(-1, 4)
    cbytes = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
    cb = np.zeros(((n + 3) // 4 + 15) // 16 * 16, dtype=np.uint8)
    cb[:cbytes.shape[0]] = cbytes
    nw = _WIRE_WORDS[rw][cls]
    keep = np.arange(rw - 1)[None, :] < nw[:, None]                      # words [1 .. nw] of every record, row-major
    words = rec[:, 1:][keep]
    return np.concatenate((cb, words.astype(np.uint32).view(np.uint8)))


def unpack_records(wire: np.ndarray, n: int, rw: int) -> np.ndarray:
    """The inverse: rec[n, rw] with word [0] = the record's place in the slice, unused words zero."""
    wire = np.ascontiguousarray(wire, dtype=np.uint8)
    nb = ((n + 3) // 4 + 15) // 16 * 16
    cb = wire[:nb]
    cls = ((cb[:, None] >> (2 * np.arange(4, dtype=np.uint8))[None, :]) & 3).reshape(-1)[:n]
    nw = _WIRE_WORDS[rw][cls]
    words = wire[nb:].view(np.uint32)
    if int(nw.sum()) != words.shape[0]:
        raise ValueError("unpack_records: the classes announce %d words, the buffer holds %d" % (int(nw.sum()), words.shape[0]))
    rec = np.zeros((n, rw), dtype=np.uint32)
    rec[:, 0] = np.arange(n, dtype=np.uint32)
    keep = np.arange(rw - 1)[None, :] < nw[:, None]
    body = rec[:, 1:]
    body[keep] = words
    rec[:, 1:] = body
    return rec
