"""ctypes binding of the C ABI in include/kasa_hip.h (libkasa_hip.so).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, calls raise.  The
reference throws std::runtime_error and main prints "ERROR: <what>" (source/main.cpp:1717-1720); the
binding turns every non-zero status into RuntimeError(kasa_last_error()) the same way.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KASA_LIB: another build of the same library (tools/asan_run.sh: the host half compiled with -fsanitize=address,undefined)
SO_PATH = os.environ.get("KASA_LIB") or os.path.join(_HERE, "libkasa_hip.so")
_lib = None

STAGES = ("encode", "sort", "lookup", "group", "regroup", "score")

EXPORTS = [
    "kasa_last_error", "kasa_device_count", "kasa_index_create", "kasa_index_destroy", "kasa_index_size",
    "kasa_index_device_bytes", "kasa_builtin_codon_table", "kasa_ctx_create", "kasa_ctx_set_protein", "kasa_ctx_destroy", "kasa_batch_upload", "kasa_batch_upload_device", "kasa_batch_upload_segments", "kasa_batch_encode",
    "kasa_batch_sort_and_range", "kasa_batch_lookup_score", "kasa_batch_group", "kasa_batch_group_to", "kasa_batch_score", "kasa_batch_records_size",
    "kasa_batch_records_fetch", "kasa_batch_records_import", "kasa_batch_scores_size", "kasa_batch_scores_fetch",
    "kasa_profile_reset", "kasa_profile_absorb", "kasa_profile_fetch", "kasa_profile_export_limbs", "kasa_profile_import_limbs", "kasa_profile_allreduce",
    "kasa_ctx_stage_ms", "kasa_ctx_stage_reset", "kasa_ctx_kernel_ms", "kasa_ctx_batch_stats", "kasa_batch_query_count", "kasa_batch_fetch_queries",
    "kasa_batch_fetch_lookup", "kasa_ctx_device_bytes", "kasa_device_memory", "kasa_batch_bytes_per_query", "kasa_ctx_counters", "kasa_ctx_third_pass_reads", "kasa_ctx_synchronize", "kasa_batch_set_queries", "kasa_ctx_debug",
    "kasa_refbatch_budget", "kasa_refbatch_sequence_cost", "kasa_refbatch_read_overhead", "kasa_refbatch_cut",
    "kasa_batch_rank", "kasa_batch_rank_fetch", "kasa_host_alloc", "kasa_host_free", "kasa_thread_device",
    "kasa_batch_queries_device", "kasa_batch_slice_starts", "kasa_batch_set_sorted_device", "kasa_batch_records_device",
    "kasa_batch_records_import_device", "kasa_batch_records_inbox", "kasa_batch_coherence",
    "kasa_ctx_set_taxa_text", "kasa_batch_text", "kasa_batch_text_fetch", "kasa_batch_text_fetch_range", "kasa_text_dtoa", "kasa_ctx_reserve", "kasa_runtime_versions", "kasa_ctx_group_tiles", "kasa_ctx_dense_reads", "kasa_ctx_replay_stats", "kasa_ctx_group_second_chance", "kasa_ctx_record_placement",
    "kasa_device_alloc", "kasa_device_free", "kasa_device_write", "kasa_device_read", "kasa_batch_records_pack_size", "kasa_batch_records_pack", "kasa_batch_records_unpack",
]


_share_torch = False
_runtime_from = "ROCm (this library's own link)"


def share_torch_runtime():
    """Opt in, BEFORE the library is loaded: this process will import torch later (bench.py, dist.py, a test that keeps reads
    in torch tensors).  A torch wheel brings its own libamdhip64.so (same SONAME as ROCm's); imported after ROCm's copy it
    would be a second HIP runtime in the process and torch would find no device.  With this call torch's copy is loaded first
    -- without importing torch -- and libkasa_hip.so's libamdhip64.so.7 resolves to it: one runtime.  A process that never
    imports torch (the C++ driver, __graft_entry__.smoke, tools/fuzz_gpu.py) does NOT do this and runs on the runtime the
    library was built for.  A process that has imported torch already needs nothing: its runtime is the one in the process.
    Either way lib() compares the version the library was built with against the one it runs on (runtime_info())."""
    global _share_torch
    if _lib is not None and not _share_torch and "torch" not in __import__("sys").modules:
        raise RuntimeError("capi.share_torch_runtime() must be called before the library is loaded (capi.lib())")
    _share_torch = True


def _one_hip_runtime():
    """Which HIP runtime libkasa_hip.so will run on (see share_torch_runtime): torch's when torch is in the process or was
    asked for (call, or KASA_TORCH_HIP_RUNTIME=1), else ROCm's own."""
    import sys
    global _runtime_from
    if "torch" in sys.modules:
        _runtime_from = "torch wheel (torch was imported first)"
        return
    if not (_share_torch or os.environ.get("KASA_TORCH_HIP_RUNTIME") == "1"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            _runtime_from = "torch wheel (share_torch_runtime)"
    except (OSError, ImportError, ValueError):
        pass


def _hip_version_text(v: int) -> str:
    return "%d.%d.%d" % (v // 10_000_000, (v // 100_000) % 100, v % 100_000) if v else "none"


def runtime_info() -> dict:
    """What the library was built with and what it runs on (kasa_runtime_versions) + the shared objects really mapped."""
    L = lib()
    v = [C.c_int(0) for _ in range(5)]
    L.kasa_runtime_versions(*[C.byref(x) for x in v])
    hb, hr, hd, rb, rr = [x.value for x in v]
    mapped = {}
    try:
        for line in open("/proc/self/maps"):
            path = line.split()[-1]
            base = os.path.basename(path)
            for key in ("libamdhip64", "librccl", "libhsa-runtime64"):
                if base.startswith(key):
                    mapped.setdefault(key, set()).add(path)
    except OSError:
        pass
    return {"hip_built": _hip_version_text(hb), "hip_runtime": _hip_version_text(hr), "hip_driver": _hip_version_text(hd),
            "hip_runtime_matches_build": (hb // 100_000) == (hr // 100_000), "runtime_from": _runtime_from,
            "rccl_built": rb, "rccl_runtime": rr, "mapped": {k: sorted(x) for k, x in mapped.items()}}


def _check_runtime(L):
    """One HIP runtime in the process, and the one the library was built for -- or the host is told.  A different MINOR
    version under the same SONAME (ROCm 7.2's hipcc over a torch wheel's 7.0 runtime) is announced once on stderr and refused
    with KASA_STRICT_HIP_RUNTIME=1; two copies of libamdhip64 in one process are always an error."""
    import sys
    hb, hr = C.c_int(0), C.c_int(0)
    L.kasa_runtime_versions(C.byref(hb), C.byref(hr), None, None, None)
    copies = set()
    try:
        for line in open("/proc/self/maps"):
            path = line.split()[-1]
            if os.path.basename(path).startswith("libamdhip64"):
                copies.add(os.path.realpath(path))
    except OSError:
        pass
    if len(copies) > 1:
        raise RuntimeError("two HIP runtimes in this process (%s): import torch before kasa_amd.capi loads its library, or call "
                           "capi.share_torch_runtime() first" % ", ".join(sorted(copies)))
    if hr.value and (hb.value // 100_000) != (hr.value // 100_000):
        msg = ("kasa_amd: libkasa_hip.so was built with HIP %s and runs on HIP runtime %s (%s)"
               % (_hip_version_text(hb.value), _hip_version_text(hr.value), _runtime_from))
        if os.environ.get("KASA_STRICT_HIP_RUNTIME") == "1":
            raise RuntimeError(msg + "; KASA_STRICT_HIP_RUNTIME=1 refuses that")
        if os.environ.get("KASA_QUIET_HIP_RUNTIME") != "1":
            sys.stderr.write(msg + "\n")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: build it with `python -m kasa_amd.build` "
                               "(there is no CPU fallback for the identify path)")
        _one_hip_runtime()
        L = C.CDLL(SO_PATH)
        L.kasa_last_error.restype = C.c_char_p
        L.kasa_index_size.restype = C.c_uint64
        L.kasa_batch_bytes_per_query.restype = C.c_uint64
        L.kasa_batch_bytes_per_query.argtypes = [C.c_void_p]
        L.kasa_index_device_bytes.restype = C.c_uint64
        L.kasa_index_size.argtypes = [C.c_void_p]
        L.kasa_index_device_bytes.argtypes = [C.c_void_p]
        L.kasa_index_destroy.argtypes = [C.c_void_p]
        L.kasa_index_destroy.restype = None
        L.kasa_ctx_destroy.argtypes = [C.c_void_p]
        L.kasa_ctx_destroy.restype = None
        L.kasa_refbatch_sequence_cost.restype = C.c_int64
        L.kasa_refbatch_sequence_cost.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int]
        L.kasa_refbatch_read_overhead.restype = C.c_int64
        L.kasa_refbatch_read_overhead.argtypes = [C.c_int64, C.c_uint32, C.c_int]
        L.kasa_refbatch_cut.restype = C.c_uint64
        L.kasa_refbatch_cut.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_uint64]
        L.kasa_host_alloc.restype = C.c_void_p
        L.kasa_host_alloc.argtypes = [C.c_size_t]
        L.kasa_host_free.restype = None
        L.kasa_host_free.argtypes = [C.c_void_p]
        L.kasa_batch_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]
        L.kasa_batch_rank_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kasa_runtime_versions.argtypes = [C.c_void_p] * 5
        _check_runtime(L)
        _lib = L
    return _lib


def builtin_codon_table() -> np.ndarray:
    lut = np.zeros(366, dtype=np.uint8)
    _check(lib().kasa_builtin_codon_table(lut.ctypes.data_as(C.c_void_p)))
    return lut


def codon_table_from_gcprt(path: str, table_id: str) -> np.ndarray:
    """-a/--alphabet <gc.prt> <id> (kASA::setCodonTable, kASA.hpp:579-615): the 64 codons of NCBI table `id` written
    over the built-in table ('*' becomes '['); codons with X or Z keep their built-in letters.  Unknown id: a warning
    and the built-in table, as in the reference."""
    import sys
    lut = builtin_codon_table()
    with open(path) as f:
        lines = f.read().split("\n")
    for i, line in enumerate(lines):
        if ("  id " + str(table_id) + " ,") in line:
            aa, b1, b2, b3 = lines[i + 1], lines[i + 3], lines[i + 4], lines[i + 5]
            pa = aa.index('"') + 1
            pb = min(x for x in (b1.find(c) for c in "TGCA") if x >= 0)
            while pb < len(b1):
                idx = ((ord(b1[pb]) & 14) << 5) | ((ord(b2[pb]) & 14) << 2) | ((ord(b3[pb]) & 14) >> 1)
                ch = aa[pa]
                lut[idx] = ord("[" if ch == "*" else ch) & 31
                pb += 1
                pa += 1
            return lut
    sys.stderr.write("WARNING: codon table not found in file. Using built-in.\n")
    return lut


def _check(rc: int):
    if rc != 0:
        raise RuntimeError(lib().kasa_last_error().decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _RefBatchParams(C.Structure):
    _fields_ = [("memoryGiB", C.c_int64), ("threads", C.c_int), ("ram", C.c_int), ("kHigh", C.c_int), ("kLow", C.c_int),
                ("recordBytes", C.c_int), ("nRecords", C.c_uint64), ("triePrefix", C.c_void_p), ("nTrie", C.c_uint64),
                ("taxIds", C.c_void_p), ("nTaxa", C.c_uint32), ("nameBytes", C.c_uint64), ("identifyMultiple", C.c_int)]


class RefBatcher:
    """Where `kASA identify -m <GiB>` cuts its batches (kasa_refbatch_* of the C ABI; host arithmetic only, works
    without a GPU).  Per-read scores are float sums whose order depends on the reads sharing a batch, so byte-identical
    per-read files need the reference's boundaries (Compare.hpp:2803-2818,3129-3132; Read.hpp:612-630,1147,1165-1195)."""

    def __init__(self, ix, k_high: int, k_low: int, frames: int = 3, memory_gib: int = 5, threads: int = 1,
                 ram: bool = False, record_bytes: int = None, identify_multiple: bool = False, coherence: bool = False):
        self.coherence = int(bool(coherence))
        tp = np.ascontiguousarray(ix.trie_prefix, dtype=np.uint32)
        tax = np.ascontiguousarray(ix.content.taxids, dtype=np.uint32)
        name_bytes = sum(len(n.replace(",", "").encode("latin-1", "replace")) for n in ix.content.names[1:])
        rb = record_bytes if record_bytes is not None else (20 if ix.K > 12 else 12)
        prm = _RefBatchParams(int(memory_gib), int(threads), int(bool(ram)), max(k_high, k_low), min(k_high, k_low), rb, int(ix.n),
                              tp.ctypes.data, len(tp), tax.ctypes.data, len(tax), name_bytes, int(bool(identify_multiple)))
        b = C.c_int64(0)
        _check(lib().kasa_refbatch_budget(C.byref(prm), C.byref(b)))
        self.budget = b.value
        self.K, self.k_low, self.frames, self.n_taxa = int(ix.K), min(k_high, k_low), frames, len(tax)

    def costs(self, reads, want_per_read: bool = True) -> np.ndarray:
        """Budget bytes per read of a ReadBatch (paired-end: both mates)."""
        L = lib()
        protein = bool(reads.protein)
        mode = 2 if protein else (1 if self.frames == 1 else 0)
        strands = 2 if (self.frames == 6 and not protein) else 1
        seq_len = np.diff(reads.offsets).astype(np.int64)
        per_seq = np.array([L.kasa_refbatch_sequence_cost(self.K, self.k_low, mode, strands, int(x), self.coherence) for x in seq_len], dtype=np.int64)
        if reads.seg_read is not None:
            cost = np.zeros(reads.n, dtype=np.int64)
            np.add.at(cost, reads.seg_read.astype(np.int64), per_seq)
        else:
            cost = per_seq
        if want_per_read:
            cost = cost + np.array([L.kasa_refbatch_read_overhead(len(n.encode("latin-1", "replace")), self.n_taxa, self.coherence) for n in reads.names], dtype=np.int64)
        return np.ascontiguousarray(cost)

    def boundaries(self, reads, want_per_read: bool = True) -> list:
        """[0, b1, b2, ..., n]: read indices where the reference starts a new batch."""
        cost = self.costs(reads, want_per_read)
        out, done, first = [0], 0, 1
        while done < len(cost):
            n = int(lib().kasa_refbatch_cut(self.budget, first, cost[done:].ctypes.data, len(cost) - done))
            if n == 0:
                n = 1          # the reference would spin on a budget below 100 MiB; always make progress
            done += n
            out.append(done)
            first = 0
        return out


    def piece_batches(self, pieced, want_per_read: bool = True) -> list:
        """The reference's batches over an input with sequences read in PIECES (ReadBatch.with_pieces): [0, p1, p2, ..., nPieces],
        piece indices.  A piece takes its k-mers and text from the budget when it is read, the read's own share (specifier,
        score row) goes with its last piece (Read.hpp:1157-1194); a batch may end between two pieces of one read."""
        L = lib()
        protein = bool(pieced.protein)
        mode = 2 if protein else (1 if self.frames == 1 else 0)
        strands = 2 if (self.frames == 6 and not protein) else 1
        seq_len = np.diff(pieced.offsets).astype(np.int64)
        cost = np.array([L.kasa_refbatch_sequence_cost(self.K, self.k_low, mode, strands, int(x), self.coherence) for x in seq_len], dtype=np.int64)
        seg = pieced.seg_read.astype(np.int64)
        last = np.ones(len(seg), bool)
        last[:-1] = seg[1:] != seg[:-1]
        if want_per_read:
            over = np.array([L.kasa_refbatch_read_overhead(len(n.encode("latin-1", "replace")), self.n_taxa, self.coherence) for n in pieced.names], dtype=np.int64)
            cost[last] += over[seg[last]]
        cost = np.ascontiguousarray(cost)
        out, done, first = [0], 0, 1
        while done < len(cost):
            n = int(L.kasa_refbatch_cut(self.budget, first, cost[done:].ctypes.data, len(cost) - done))
            done += max(n, 1)
            out.append(done)
            first = 0
        return out


RANK_ENTRY = np.dtype([("tax", np.uint32), ("score", np.float32), ("rel", np.float64)])


class TextParams(C.Structure):
    """kasa_text_params (include/kasa_hip.h)"""
    _fields_ = [("format", C.c_int), ("beasts", C.c_uint32), ("firstRead", C.c_uint64), ("readNames", C.c_char_p),
                ("readNameOff", C.POINTER(C.c_uint64)), ("readLen", C.POINTER(C.c_uint32)), ("bestScore", C.POINTER(C.c_float)),
                ("nClasses", C.c_uint32), ("coherence", C.c_int), ("errorThreshold", C.c_double), ("coherenceThreshold", C.c_float)]


def device_dtoa(values, device: int = 0):
    """The reference's double -> text as the DEVICE writes it (kasa_text_dtoa): list of str."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.zeros(v.shape[0] * 32, dtype=np.uint8)
    _check(lib().kasa_text_dtoa(C.c_int(device), _p(v), C.c_uint32(v.shape[0]), _p(out)))
    return [bytes(out[i * 32:(i + 1) * 32]).split(b"\0", 1)[0].decode("ascii") for i in range(v.shape[0])]


def pinned_empty(n: int, dtype) -> np.ndarray:
    """numpy array over page-locked memory (kasa_host_alloc): PCIe transfers to and from it run at link rate.  The block
    is freed when the array (and every view of it) is gone."""
    import weakref
    dtype = np.dtype(dtype)
    nbytes = max(1, int(n) * dtype.itemsize)
    ptr = lib().kasa_host_alloc(C.c_size_t(nbytes))
    if not ptr:
        raise MemoryError(f"kasa_host_alloc({nbytes}) failed")
    arr = np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype, count=int(n))
    weakref.finalize(arr, lib().kasa_host_free, C.c_void_p(ptr))
    return arr


class DeviceBuffer:
    """Plain device memory through the C ABI (kasa_device_alloc / _write / _free): for hosts that keep inputs resident in HBM
    without importing another GPU library (bench.py at N = 1).  `ptr` is the device address."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = int(device), int(nbytes)
        p = C.c_void_p(0)
        _check(lib().kasa_device_alloc(C.c_int(self.device), C.c_size_t(self.nbytes), C.byref(p)))
        self.ptr = int(p.value or 0)

    def write(self, arr: np.ndarray, byte_offset: int = 0):
        arr = np.ascontiguousarray(arr)
        if byte_offset + arr.nbytes > self.nbytes:
            raise ValueError("DeviceBuffer.write beyond the buffer")
        _check(lib().kasa_device_write(C.c_int(self.device), C.c_void_p(self.ptr + int(byte_offset)), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))

    def read(self, nbytes: int = None, byte_offset: int = 0) -> np.ndarray:
        nbytes = self.nbytes - byte_offset if nbytes is None else int(nbytes)
        out = np.empty(nbytes, dtype=np.uint8)
        _check(lib().kasa_device_read(C.c_int(self.device), out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + int(byte_offset)), C.c_size_t(nbytes)))
        return out

    def close(self):
        if self.ptr:
            _check(lib().kasa_device_free(C.c_int(self.device), C.c_void_p(self.ptr)))
            self.ptr = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_memory(device: int = 0):
    """(free, total) bytes of HBM (hipMemGetInfo)."""
    free, total = C.c_uint64(0), C.c_uint64(0)
    _check(lib().kasa_device_memory(C.c_int(device), C.byref(free), C.byref(total)))
    return int(free.value), int(total.value)


def bytes_per_query(ctx) -> int:
    """Device bytes one query k-mer of a batch takes in this context (kasa_batch_bytes_per_query)."""
    return int(lib().kasa_batch_bytes_per_query(ctx.h))


def device_count() -> int:
    n = C.c_int(0)
    lib().kasa_device_count(C.byref(n))
    return n.value


class DeviceIndex:
    """HBM-resident index (immutable, shareable between contexts)."""

    def __init__(self, ix, device: int = 0, check_trie: bool = True):
        from .formats import REC_DTYPE, REC128_DTYPE, is_wide
        self.wide = bool(is_wide(ix.kmer))
        if self.wide:   # 128-bit index: 20-byte {low, high, taxid} records (packedLargePair)
            rec = np.zeros(ix.n, dtype=REC128_DTYPE)
            rec["lo"], rec["hi"], rec["tax"] = ix.kmer["lo"], ix.kmer["hi"], ix.taxid
        else:
            rec = np.zeros(ix.n, dtype=REC_DTYPE)
            rec["kmer"], rec["tax"] = ix.kmer, ix.taxid
        tp = np.ascontiguousarray(ix.trie_prefix, dtype=np.uint32) if check_trie else None
        tc = np.ascontiguousarray(ix.trie_count, dtype=np.uint64) if check_trie else None
        ids = np.ascontiguousarray(ix.content.taxids, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().kasa_index_create(C.c_int(device), _p(rec), C.c_uint64(ix.n), C.c_int(rec.dtype.itemsize), _p(tp), _p(tc),
                                       C.c_uint64(0 if tp is None else tp.shape[0]), _p(ids),
                                       C.c_uint32(ids.shape[0]), C.byref(h)))
        self.h = h
        self.device = device
        self.n_taxa = int(ids.shape[0])
        self.n = ix.n

    @classmethod
    def from_device_records(cls, records_ptr: int, n: int, record_bytes: int, taxids, device: int = 0):
        """An index whose file-layout records ({kmer, taxid}, 12 or 20 bytes each, sorted, unique) already lie in device
        memory (bench.py synthesises the slices of a range-partitioned index there); no `_trie` cross-check."""
        self = cls.__new__(cls)
        ids = np.ascontiguousarray(taxids, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().kasa_index_create(C.c_int(device), C.c_void_p(records_ptr), C.c_uint64(n), C.c_int(record_bytes), None, None,
                                       C.c_uint64(0), _p(ids), C.c_uint32(ids.shape[0]), C.byref(h)))
        self.h, self.device, self.n_taxa, self.n, self.wide = h, device, int(ids.shape[0]), int(n), record_bytes == 20
        return self

    @property
    def device_bytes(self) -> int:
        return int(lib().kasa_index_device_bytes(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib().kasa_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One identify context: stream, batch buffers, profile tables (single-threaded)."""

    def __init__(self, dix: DeviceIndex, k_high=12, k_low=7, frames=3, codon_lut=None):
        self.dix = dix
        self.k_high, self.k_low = max(k_high, k_low), min(k_high, k_low)
        self.nK = self.k_high - self.k_low + 1
        self.n_reads = 0
        self.n_kmers = 0
        h = C.c_void_p()
        lut = np.ascontiguousarray(codon_lut, dtype=np.uint8) if codon_lut is not None else None
        _check(lib().kasa_ctx_create(dix.h, C.c_int(k_high), C.c_int(k_low), C.c_int(frames), _p(lut), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().kasa_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- batch ----
    def upload(self, bases: np.ndarray, offsets: np.ndarray, seg_read: np.ndarray = None, n_reads: int = None):
        """One sequence per read, or -- paired-end -- `seg_read[s]` = read of sequence s (ascending) and `n_reads`."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n_seq = int(offsets.shape[0] - 1)
        if seg_read is None:
            self.n_reads = n_seq
            _check(lib().kasa_batch_upload(self.h, _p(bases), _p(offsets), C.c_int64(n_seq)))
        else:
            seg_read = np.ascontiguousarray(seg_read, dtype=np.uint32)
            self.n_reads = int(n_reads)
            _check(lib().kasa_batch_upload_segments(self.h, _p(bases), _p(offsets), C.c_int64(n_seq), _p(seg_read),
                                                    C.c_int64(self.n_reads)))

    def upload_device(self, bases_ptr: int, offsets: np.ndarray):
        """upload() of reads whose bases already lie in device memory at `bases_ptr` (offsets: host, relative to it)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n_reads = int(offsets.shape[0] - 1)
        _check(lib().kasa_batch_upload(self.h, C.c_void_p(bases_ptr), _p(offsets), C.c_int64(self.n_reads)))

    def upload_resident(self, bases_ptr: int, offsets_ptr: int, n_reads: int):
        """kasa_batch_upload_device: bases AND offsets (int64[n_reads + 1]) lie in device memory and are read in place --
        nothing crosses PCIe; the caller keeps both alive until the batch is done."""
        self.n_reads = int(n_reads)
        _check(lib().kasa_batch_upload_device(self.h, C.c_void_p(bases_ptr), C.c_void_p(offsets_ptr), C.c_int64(self.n_reads)))

    def profile_allreduce(self, comm: int):
        """kasa_profile_allreduce: the profile tables summed over the ranks of an RCCL communicator (ncclComm_t as an
        integer), on the context's stream, no host copy."""
        _check(lib().kasa_profile_allreduce(self.h, C.c_void_p(comm)))

    def encode(self) -> int:
        n = C.c_uint64(0)
        _check(lib().kasa_batch_encode(self.h, C.byref(n)))
        self.n_kmers = int(n.value)
        return self.n_kmers

    def set_protein(self, protein: bool):
        """Amino-acid input for the next batches (kASA.hpp:155-183)."""
        _check(lib().kasa_ctx_set_protein(self.h, C.c_int(int(protein))))

    def sort_and_range(self, unique: bool = False):
        _check(lib().kasa_batch_sort_and_range(self.h, C.c_int(int(unique))))

    def lookup_score(self, want_per_read: bool = True, coverage: bool = False):
        _check(lib().kasa_batch_lookup_score(self.h, C.c_int(int(want_per_read)), C.c_int(int(coverage))))

    def group(self, coverage: bool = False):
        _check(lib().kasa_batch_group(self.h, C.c_int(int(coverage))))

    def group_to(self, records_ptr: int, coverage: bool = False):
        """group() with the records written straight into device memory at `records_ptr` (kasa_batch_group_to)."""
        _check(lib().kasa_batch_group_to(self.h, C.c_int(int(coverage)), C.c_void_p(records_ptr)))

    def score(self, want_per_read: bool = True):
        _check(lib().kasa_batch_score(self.h, C.c_int(int(want_per_read))))

    @property
    def rec_words(self) -> int:
        """32-bit words of one event record (include/kasa_hip.h, kasa_batch_group): 8 up to 8 levels, 16 up to 25."""
        return 8 if self.nK <= 8 else 16

    def records(self):
        """Event records of the batch after group(): (rec u32[nQ, rec_words], pool u32[]), records in sorted order."""
        n, m = C.c_uint64(0), C.c_uint64(0)
        _check(lib().kasa_batch_records_size(self.h, C.byref(n), C.byref(m)))
        rec = np.zeros(int(n.value), dtype=np.uint32)
        pool = np.zeros(max(1, int(m.value)), dtype=np.uint32)
        _check(lib().kasa_batch_records_fetch(self.h, _p(rec), _p(pool)))
        return rec.reshape(-1, self.rec_words), pool

    def records_import(self, rec: np.ndarray, pool: np.ndarray):
        rec = np.ascontiguousarray(rec, dtype=np.uint32).reshape(-1)
        pool = np.ascontiguousarray(pool, dtype=np.uint32)
        _check(lib().kasa_batch_records_import(self.h, _p(rec), C.c_uint64(rec.shape[0]), _p(pool), C.c_uint64(pool.shape[0])))

    # ---- device-resident exchange (C5): device pointers as plain integers
    def queries_device(self):
        """(device pointer of the sorted k-mers, number of queries, bytes per k-mer)."""
        p, n = C.c_void_p(0), C.c_uint64(0)
        _check(lib().kasa_batch_queries_device(self.h, C.byref(p), C.byref(n)))
        return int(p.value or 0), int(n.value), (16 if self.dix.wide else 8)

    def slice_starts(self, cuts: np.ndarray) -> np.ndarray:
        cuts = np.ascontiguousarray(cuts, dtype=np.uint64)
        starts = np.zeros(cuts.shape[0] + 1, dtype=np.uint64)
        _check(lib().kasa_batch_slice_starts(self.h, _p(cuts), C.c_uint32(cuts.shape[0]), _p(starts)))
        return starts.astype(np.int64)

    def set_sorted_device(self, ptr: int, n: int):
        _check(lib().kasa_batch_set_sorted_device(self.h, C.c_void_p(ptr), C.c_uint64(n)))
        self.n_kmers = int(n)

    def records_device(self):
        """(records pointer, record words, pool pointer, pool words) of the grouped slice, all on the device."""
        r, p, nr, npw = C.c_void_p(0), C.c_void_p(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().kasa_batch_records_device(self.h, C.byref(r), C.byref(nr), C.byref(p), C.byref(npw)))
        return int(r.value or 0), int(nr.value), int(p.value or 0), int(npw.value)

    def records_inbox(self, n_record_words: int) -> int:
        """Device pointer of the context's staging buffer for imported records (room for n_record_words u32): records
        received there are shifted and filed in place by records_import_device."""
        p = C.c_void_p(0)
        _check(lib().kasa_batch_records_inbox(self.h, C.c_uint64(n_record_words), C.byref(p)))
        return int(p.value or 0)

    def records_pack_size(self, records_ptr: int, n_queries: int) -> int:
        """Bytes the exported records at `records_ptr` (device) take on the wire (kasa_batch_records_pack_size)."""
        nb = C.c_uint64(0)
        _check(lib().kasa_batch_records_pack_size(self.h, C.c_void_p(records_ptr), C.c_uint64(n_queries), C.byref(nb)))
        return int(nb.value)

    def records_pack(self, records_ptr: int, n_queries: int, out_ptr: int, cap_bytes: int):
        _check(lib().kasa_batch_records_pack(self.h, C.c_void_p(records_ptr), C.c_uint64(n_queries), C.c_void_p(out_ptr), C.c_uint64(cap_bytes)))

    def records_unpack(self, packed_ptr: int, n_bytes: int, n_queries: int, out_ptr: int):
        _check(lib().kasa_batch_records_unpack(self.h, C.c_void_p(packed_ptr), C.c_uint64(n_bytes), C.c_uint64(n_queries), C.c_void_p(out_ptr)))

    def records_import_device(self, parts):
        """parts[j] = (records pointer, record words, pool pointer, pool words) of slice j, in partition order."""
        n = len(parts)
        rp = (C.c_void_p * n)(*[C.c_void_p(a[0]) for a in parts])
        pp = (C.c_void_p * n)(*[C.c_void_p(a[2]) for a in parts])
        rw = (C.c_uint64 * n)(*[a[1] for a in parts])
        pw = (C.c_uint64 * n)(*[a[3] for a in parts])
        _check(lib().kasa_batch_records_import_device(self.h, C.c_uint32(n), rp, rw, pp, pw))

    def scores(self, pinned: bool = False, out=None):
        """CSR (offsets u64[nReads+1], taxIdx u32[nnz], score f32[nnz]); pinned: into page-locked memory (link rate);
        out = (off, tax, sc): buffers of a long-lived host (page-locking 10 GB takes about a second), at least as large."""
        nnz = C.c_uint64(0)
        _check(lib().kasa_batch_scores_size(self.h, C.byref(nnz)))
        if out is not None:
            off, tax, sc = out[0][:self.n_reads + 1], out[1][:nnz.value], out[2][:nnz.value]
            assert off.shape[0] == self.n_reads + 1 and tax.shape[0] == nnz.value and sc.shape[0] == nnz.value
            _check(lib().kasa_batch_scores_fetch(self.h, _p(off), _p(tax), _p(sc)))
            return off, tax, sc
        alloc = pinned_empty if pinned else (lambda n, dt: np.zeros(n, dtype=dt))
        off = alloc(self.n_reads + 1, np.uint64)
        tax = alloc(nnz.value, np.uint32)
        sc = alloc(nnz.value, np.float32)
        _check(lib().kasa_batch_scores_fetch(self.h, _p(off), _p(tax), _p(sc)))
        return off, tax, sc

    def rank(self, den: np.ndarray, read_class: np.ndarray, threshold: float, beasts: int, pinned: bool = True, out=None):
        """kasa_batch_rank + fetch: (meta u32[nReads, 4], entries RANK_ENTRY[nEntries], nFlagged).  den: float64
        [nClasses, nTaxa]; read_class: uint32[nReads]; out = (meta u32[4 nReads], entries) buffers to reuse."""
        den = np.ascontiguousarray(den, dtype=np.float64)
        read_class = np.ascontiguousarray(read_class, dtype=np.uint32)
        n_ent, n_flag = C.c_uint64(0), C.c_uint32(0)
        _check(lib().kasa_batch_rank(self.h, _p(den), C.c_uint32(den.shape[0]), _p(read_class), C.c_float(threshold),
                                     C.c_uint32(beasts), C.byref(n_ent), C.byref(n_flag)))
        alloc = pinned_empty if pinned else (lambda n, dt: np.empty(n, dtype=dt))
        if out is not None and out[1].shape[0] >= n_ent.value:
            meta, ent = out[0][:self.n_reads * 4], out[1][:n_ent.value]
        else:
            meta = alloc(self.n_reads * 4, np.uint32)
            ent = alloc(n_ent.value, RANK_ENTRY)
        _check(lib().kasa_batch_rank_fetch(self.h, _p(meta), _p(ent)))
        return meta.reshape(-1, 4), ent, n_flag.value

    def set_taxa_text(self, taxids, names):
        """kasa_ctx_set_taxa_text: what the device prints for a taxon (content file: entry 0 = non_unique)."""
        ids = np.ascontiguousarray(taxids, dtype=np.uint32)
        enc = [n.encode("latin-1") if isinstance(n, str) else bytes(n) for n in names]
        if len(enc) != self.dix.n_taxa or ids.shape[0] != self.dix.n_taxa:
            raise ValueError("set_taxa_text: one id and one name per taxon of the index")
        off = np.zeros(len(enc) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
        blob = b"".join(enc)
        _check(lib().kasa_ctx_set_taxa_text(self.h, _p(ids), C.c_char_p(blob), _p(off)))

    TEXT_FORMATS = {"tsv": 0, "json": 1, "jsonl": 2, "kraken": 3}

    def text(self, fmt: str, beasts: int, first_read: int, names, lengths, best, coherence: bool = False,
             error_threshold: float = 0.5, coherence_threshold: float = 11.0, pinned: bool = False, pieces: int = 1):
        """kasa_batch_text + fetch: the per-read file's bytes of the batch last ranked, written on the device.
        names: the specifiers as printed; lengths: uint32[nReads]; best: float32[nClasses] (the classes given to rank()).
        -> (text bytes, uint64 offsets[nReads + 1], uint8 contaminated[nReads])"""
        enc = [n.encode("latin-1") if isinstance(n, str) else bytes(n) for n in names]
        if len(enc) != self.n_reads:
            raise ValueError("text: one name per read")
        off = np.zeros(len(enc) + 1, dtype=np.uint64)
        if enc:
            off[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
        blob = b"".join(enc)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        best = np.ascontiguousarray(best, dtype=np.float32)
        tp = TextParams(self.TEXT_FORMATS[fmt], int(beasts), int(first_read), C.c_char_p(blob), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                        lengths.ctypes.data_as(C.POINTER(C.c_uint32)), best.ctypes.data_as(C.POINTER(C.c_float)), int(best.shape[0]),
                        1 if coherence else 0, float(np.float32(error_threshold)), float(np.float32(coherence_threshold)))
        n = C.c_uint64(0)
        _check(lib().kasa_batch_text(self.h, C.byref(tp), C.byref(n)))
        buf = (pinned_empty if pinned else (lambda k, dt: np.empty(k, dtype=dt)))(max(1, n.value), np.uint8)
        offs = np.zeros(self.n_reads + 1, dtype=np.uint64)
        flags = np.zeros(max(1, self.n_reads), dtype=np.uint8)
        if pieces <= 1:
            _check(lib().kasa_batch_text_fetch(self.h, _p(buf), _p(offs), _p(flags)))
        else:                                                        # the text in pieces (kasa_batch_text_fetch_range), the rest as before
            _check(lib().kasa_batch_text_fetch(self.h, None, _p(offs), _p(flags)))
            step = max(1, (n.value + pieces - 1) // pieces)
            for a in range(0, n.value, step):
                k = min(step, n.value - a)
                _check(lib().kasa_batch_text_fetch_range(self.h, C.c_void_p(buf.ctypes.data + a), C.c_uint64(a), C.c_uint64(k)))
        return buf[:n.value].tobytes(), offs, flags[:self.n_reads]

    def coherence(self) -> np.ndarray:
        """--coherence scores of the batch (Compare::postProcess): float32[n_reads].  Raises where the reference throws."""
        sc = np.zeros(self.n_reads, dtype=np.float32)
        at = C.c_uint64(0)
        _check(lib().kasa_batch_coherence(self.h, _p(sc), C.byref(at)))
        if at.value != 0xFFFFFFFFFFFFFFFF:
            raise RuntimeError(f"vector::_M_range_check: __n (which is {at.value}) >= this->size() (which is {at.value})")
        return sc

    def run_batch(self, bases, offsets, want_per_read=True, coverage=False, unique=False, seg_read=None, n_reads=None):
        self.upload(bases, offsets, seg_read, n_reads)
        self.encode()
        self.sort_and_range(unique)
        self.lookup_score(want_per_read, coverage)

    # ---- profile ----
    def profile_reset(self):
        _check(lib().kasa_profile_reset(self.h))

    def profile_absorb(self, other: "Context"):
        """self += other, other = 0 (kasa_profile_absorb): the tables of a context that grouped slices for this one."""
        _check(lib().kasa_profile_absorb(self.h, other.h))

    def profile(self):
        shape = (self.nK, self.dix.n_taxa)
        ca = np.zeros(shape, dtype=np.float64)
        cu = np.zeros(shape, dtype=np.uint64)
        ct = np.zeros(shape, dtype=np.uint64)
        _check(lib().kasa_profile_fetch(self.h, _p(ca), _p(cu), _p(ct)))
        return ca, cu, ct

    def profile_limbs(self) -> np.ndarray:
        out = np.zeros((self.nK * self.dix.n_taxa, 6), dtype=np.uint64)
        _check(lib().kasa_profile_export_limbs(self.h, _p(out)))
        return out

    def profile_set_limbs(self, limbs: np.ndarray):
        limbs = np.ascontiguousarray(limbs, dtype=np.uint64)
        _check(lib().kasa_profile_import_limbs(self.h, _p(limbs)))

    # ---- measurement / taps ----
    def stage_ms(self):
        out = {}
        for i, name in enumerate(STAGES):
            ms, n = C.c_double(0), C.c_uint64(0)
            _check(lib().kasa_ctx_stage_ms(self.h, C.c_int(i), C.byref(ms), C.byref(n)))
            out[name] = (ms.value, int(n.value))
        return out

    def stage_reset(self):
        _check(lib().kasa_ctx_stage_reset(self.h))

    KERNELS = ("lookup_tile_kernel", "group_kernel", "score_main_kernel", "score_other_kernel", "row_merge_kernel",
               "score_general_kernels", "profile_table_kernels", "row_copy_kernels", "sort_pass_kernels", "bucket_rank_kernel", "score_dense_kernel", "score_replay_kernels")

    def kernel_ms(self):
        """HIP-event time of single kernels alone since stage_reset(): {name: (ms, launches)}."""
        out = {}
        for i, name in enumerate(self.KERNELS):
            ms, n = C.c_double(0), C.c_uint64(0)
            _check(lib().kasa_ctx_kernel_ms(self.h, C.c_int(i), C.byref(ms), C.byref(n)))
            out[name] = (ms.value, int(n.value))
        return out

    def third_pass_reads(self) -> int:
        """Reads of the last batch whose pending window did not fit the second pass either (kasa_ctx_third_pass_reads)."""
        n = C.c_uint32(0)
        _check(lib().kasa_ctx_third_pass_reads(self.h, C.byref(n)))
        return int(n.value)

    def batch_stats(self):
        st = np.zeros(8, dtype=np.uint64)
        _check(lib().kasa_ctx_batch_stats(self.h, _p(st)))
        keys = ("queries", "staging_records", "profile_keys", "pool_words", "general_reads", "second_pass_reads", "nnz", "encoder_ranked")
        out = {k: int(v) for k, v in zip(keys, st)}
        out["group_tiles"], out["group_tiles_listed"] = self.group_tiles()
        again = C.c_uint32(0)
        _check(lib().kasa_ctx_group_second_chance(self.h, C.byref(again)))
        out["group_tiles_listed_again"] = int(again.value)
        n = C.c_uint32(0)
        _check(lib().kasa_ctx_dense_reads(self.h, C.byref(n)))
        out["dense_reads"] = int(n.value)
        ev = C.c_uint64(0)
        _check(lib().kasa_ctx_replay_stats(self.h, C.byref(n), C.byref(ev)))
        out["replay_reads"], out["replay_events"] = int(n.value), int(ev.value)
        return out

    def record_placement(self):
        """How the record buffer was chosen (kasa_ctx_record_placement): candidates timed, the kept one's rate of random 32-byte
        stores and every candidate's, in G records/s; candidates = 0: allocated plainly."""
        n, kept = C.c_uint32(0), C.c_float(0)
        rates = (C.c_float * 4)()
        _check(lib().kasa_ctx_record_placement(self.h, C.byref(n), C.byref(kept), rates))
        return {"candidates": int(n.value), "kept_g_records_per_s": round(float(kept.value), 2), "candidates_g_records_per_s": [round(float(x), 2) for x in rates[:int(n.value)]]}

    def group_tiles(self):
        """(tiles of the last batch's group stage, tiles group2_kernel left to group_kernel) -- kasa_ctx_group_tiles."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        _check(lib().kasa_ctx_group_tiles(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def max_queries_per_batch(self, device: int = 0, fraction: float = 0.8) -> int:
        """How many query k-mers fit one batch in the HBM that is free right now (at most 2^32 - 16)."""
        free, total = C.c_uint64(0), C.c_uint64(0)
        _check(lib().kasa_device_memory(C.c_int(device), C.byref(free), C.byref(total)))
        per = int(lib().kasa_batch_bytes_per_query(self.h))
        return int(min(0xFFFFFFF0 - 1, max(1 << 20, fraction * free.value / per)))

    def counters(self):
        """(reads of the last batch on the general score kernel, those of them that needed its second pass)."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        _check(lib().kasa_ctx_counters(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def query_count(self) -> int:
        n = C.c_uint64(0)
        _check(lib().kasa_batch_query_count(self.h, C.byref(n)))
        return int(n.value)

    def queries(self):
        from .formats import KEY128_DTYPE
        n = self.query_count()
        km = np.zeros(n, dtype=KEY128_DTYPE if self.dix.wide else np.uint64)
        rd = np.zeros(n, dtype=np.uint32)
        _check(lib().kasa_batch_fetch_queries(self.h, _p(km), _p(rd), C.c_uint64(n)))
        return km, rd

    def set_queries(self, kmers: np.ndarray, reads: np.ndarray, n_reads: int):
        from .formats import KEY128_DTYPE
        kmers = np.ascontiguousarray(kmers, dtype=KEY128_DTYPE if self.dix.wide else np.uint64)
        reads = np.ascontiguousarray(reads, dtype=np.uint32)
        _check(lib().kasa_batch_set_queries(self.h, _p(kmers), _p(reads), C.c_uint64(kmers.shape[0]), C.c_int64(n_reads)))
        self.n_reads, self.n_kmers = int(n_reads), int(kmers.shape[0])

    def lookup(self):
        d = np.zeros(self.n_kmers, dtype=np.uint8)
        pos = np.zeros(self.n_kmers, dtype=np.uint32)
        _check(lib().kasa_batch_fetch_lookup(self.h, _p(d), _p(pos), C.c_uint64(self.n_kmers)))
        return d, pos

    def device_bytes(self) -> int:
        b = C.c_uint64(0)
        _check(lib().kasa_ctx_device_bytes(self.h, C.byref(b)))
        return int(b.value)

    def force_slow_score(self, on: bool):
        _check(lib().kasa_ctx_debug(self.h, C.c_int(int(on)), None))

    def debug_flags(self, flags: int):
        """bit 0: general score kernel for every read; bit 1: per-query lookup instead of streamed tiles;
        bit 2: sorting row merge instead of the bitmap one."""
        _check(lib().kasa_ctx_debug(self.h, C.c_int(int(flags)), None))

    def last_slow_reads(self) -> int:
        n = C.c_uint32(0)
        _check(lib().kasa_ctx_debug(self.h, C.c_int(-1), C.byref(n)))
        return int(n.value)

    def synchronize(self):
        _check(lib().kasa_ctx_synchronize(self.h))
