"""ctypes wrapper around oracle/libkasa_oracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing in kasa_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
RANGE_NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


class Params(C.Structure):
    _fields_ = [("K", C.c_int32), ("kHigh", C.c_int32), ("kLow", C.c_int32), ("frames", C.c_int32),
                ("avxQuirk", C.c_int32), ("coverage", C.c_int32), ("protein", C.c_int32), ("cmp64Quirk", C.c_int32)]


class _Index(C.Structure):
    _fields_ = [("kmer", C.c_void_p), ("tax", C.c_void_p), ("n", C.c_uint64),
                ("triePrefix", C.c_void_p), ("trieStart", C.c_void_p), ("trieLenM1", C.c_void_p),
                ("nTrie", C.c_uint64), ("nTaxa", C.c_uint32)]


KEY128 = np.dtype([("lo", "<u8"), ("hi", "<u8")], align=False)   # unsigned __int128, little endian


def build(force: bool = False, wide: bool = False) -> str:
    so = os.path.join(_HERE, "libkasa_oracle.so")
    so128 = os.path.join(_HERE, "libkasa_oracle128.so")
    src = os.path.join(_HERE, "kasa_oracle.c")
    if force or any(not os.path.exists(x) or os.path.getmtime(x) < os.path.getmtime(src) for x in (so, so128)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so128 if wide else so


_LIBS = {}


def key_dtype(K: int):
    return KEY128 if K > 12 else np.dtype(np.uint64)


def lib(K: int = 12):
    """The 64-bit-key library (K = 12 letters) or the 128-bit-key one (K = 25), same source."""
    wide = K > 12
    if wide not in _LIBS:
        _LIB = C.CDLL(build(wide=wide))
        _LIB.ko_encode_batch.restype = C.c_int64
        _LIB.ko_unique_queries.restype = C.c_uint64
        _LIB.ko_padded_len.restype = C.c_int64
        _LIB.ko_kmer_count.restype = C.c_int64
        _LIB.ko_best_score.restype = C.c_float
        _LIB.ko_relative_score.restype = C.c_double
        _LIB.ko_error_score.restype = C.c_double
        _LIB.ko_weight.restype = C.c_float
        _LIB.ko_best_score.argtypes = [C.c_uint64, C.POINTER(Params)]
        _LIB.ko_relative_score.argtypes = [C.c_float, C.c_uint64, C.c_uint64, C.POINTER(Params)]
        _LIB.ko_error_score.argtypes = [C.c_float, C.c_float]
        _LIB.ko_padded_len.argtypes = [C.c_int64, C.POINTER(Params)]
        _LIB.ko_kmer_count.argtypes = [C.c_int64, C.POINTER(Params)]
        _LIBS[wide] = _LIB
    return _LIBS[wide]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def params(k_high=12, k_low=7, frames=3, avx_quirk=False, coverage=False, K=12, protein=False,
           cmp64_quirk=False) -> Params:
    return Params(K, k_high, k_low, frames, int(avx_quirk), int(coverage), int(protein), int(cmp64_quirk))


def codon_table() -> np.ndarray:
    lut = np.zeros(366, dtype=np.uint8)
    lib().ko_codon_table(_p(lut))
    return lut


def codon_table_from_file(path: str, table_id: str) -> np.ndarray:
    """-a <gc.prt> <id> over the built-in table (kASA.hpp:579-615)."""
    lut = codon_table()
    rc = lib().ko_codon_table_from_file(path.encode(), str(table_id).encode(), _p(lut))
    assert rc >= 0
    return lut


def encode(bases: np.ndarray, offsets: np.ndarray, p: Params, lut=None):
    """-> (kmer u64[nQ], read u32[nQ]) in emission order (Read.hpp:84-293)."""
    L = lib(p.K)
    lut = codon_table() if lut is None else lut
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.shape[0] - 1
    total = L.ko_encode_batch(_p(bases), _p(offsets), C.c_int64(n), C.byref(p), _p(lut), None, None)
    km = np.zeros(total, dtype=key_dtype(p.K))
    rd = np.zeros(total, dtype=np.uint32)
    got = L.ko_encode_batch(_p(bases), _p(offsets), C.c_int64(n), C.byref(p), _p(lut), _p(km), _p(rd))
    assert got == total
    return km, rd


def _K_of(km: np.ndarray) -> int:
    return 25 if km.dtype == KEY128 else 12


def sort_queries(km: np.ndarray, rd: np.ndarray):
    km, rd = km.copy(), rd.copy()
    lib(_K_of(km)).ko_sort_queries(_p(km), _p(rd), C.c_uint64(km.shape[0]))
    return km, rd


class IndexView:
    """Keeps the numpy arrays alive next to the C struct."""

    def __init__(self, ix):
        self.kmer = np.ascontiguousarray(ix.kmer)   # u64 or KEY128 (same layout as formats.KEY128_DTYPE)
        self.K = 25 if self.kmer.dtype.itemsize == 16 else 12
        self.tax = np.ascontiguousarray(ix.tax, dtype=np.uint32)
        self.tp = np.ascontiguousarray(ix.trie_prefix, dtype=np.uint32)
        self.ts = np.ascontiguousarray(ix.trie_start, dtype=np.uint64)
        self.tl = np.ascontiguousarray(ix.trie_len_m1, dtype=np.uint32)
        self.n_taxa = ix.content.n_taxa
        self.c = _Index(_p(self.kmer), _p(self.tax), self.kmer.shape[0], _p(self.tp), _p(self.ts),
                        _p(self.tl), self.tp.shape[0], self.n_taxa)


def ranges(iv: IndexView, p: Params, km: np.ndarray):
    rs = np.zeros(km.shape[0], dtype=np.uint64)
    rl = np.zeros(km.shape[0], dtype=np.uint32)
    lib(p.K).ko_ranges(C.byref(iv.c), C.byref(p), _p(km), C.c_uint64(km.shape[0]), _p(rs), _p(rl))
    return rs, rl


@dataclass
class CompareResult:
    count_all: np.ndarray     # f64[nK, nTaxa], row 0 = kHigh
    count_unique: np.ndarray  # u64[nK, nTaxa]
    count_total: np.ndarray   # u64[nK, nTaxa]
    M: np.ndarray             # f32[nReads, nTaxa] or None


def compare(iv: IndexView, p: Params, km, rd, rs, rl, n_reads: int, want_reads=True,
            closed_form=False) -> CompareResult:
    nK = p.kHigh - p.kLow + 1
    ca = np.zeros((nK, iv.n_taxa), dtype=np.float64)
    cu = np.zeros((nK, iv.n_taxa), dtype=np.uint64)
    ct = np.zeros((nK, iv.n_taxa), dtype=np.uint64)
    M = np.zeros((n_reads, iv.n_taxa), dtype=np.float32) if want_reads else None
    fn = lib(p.K).ko_compare_closed_form if closed_form else lib(p.K).ko_compare_sequential
    rc = fn(C.byref(p), C.byref(iv.c), _p(km), _p(rd), _p(rs), _p(rl), C.c_uint64(km.shape[0]),
            C.c_uint64(n_reads), _p(ca), _p(cu), _p(ct), _p(M) if want_reads else None)
    assert rc == 0
    return CompareResult(ca, cu, ct, M)


def compare_ml(iv: IndexView, p: Params, km, rd, rs, rl, n_reads: int, closed_form=False):
    """compare() that also returns, per sorted query, the match length setMatchLength leaves (sequential) or the deepest
    matched level (closed form): (CompareResult, u8[nQ])."""
    nK = p.kHigh - p.kLow + 1
    ca = np.zeros((nK, iv.n_taxa), dtype=np.float64)
    cu = np.zeros((nK, iv.n_taxa), dtype=np.uint64)
    ct = np.zeros((nK, iv.n_taxa), dtype=np.uint64)
    M = np.zeros((n_reads, iv.n_taxa), dtype=np.float32)
    ml = np.zeros(km.shape[0], dtype=np.uint8)
    fn = lib(p.K).ko_compare_closed_form_ml if closed_form else lib(p.K).ko_compare_sequential_ml
    rc = fn(C.byref(p), C.byref(iv.c), _p(km), _p(rd), _p(rs), _p(rl), C.c_uint64(km.shape[0]),
            C.c_uint64(n_reads), _p(ca), _p(cu), _p(ct), _p(M), _p(ml))
    assert rc == 0
    return CompareResult(ca, cu, ct, M), ml


class ReferenceThrows(Exception):
    """The reference ends with an exception on this input (its message: what())."""


def emission_geometry(offsets: np.ndarray, p: Params):
    """(read, frame, position) of every k-mer in emission order (Read.hpp:128-131: position = window number within the
    strand, frame = 0 forward / 1 reverse complement)."""
    L = lib(p.K)
    strands = 2 if (p.frames == 6 and not p.protein) else 1
    rd, fr, ps = [], [], []
    for r in range(offsets.shape[0] - 1):
        raw = int(offsets[r + 1] - offsets[r])
        cnt = int(L.ko_kmer_count(C.c_int64(L.ko_padded_len(C.c_int64(raw), C.byref(p))), C.byref(p))) if raw > 0 else 0
        for s_ in range(strands):
            rd.append(np.full(cnt, r, dtype=np.uint32)); fr.append(np.full(cnt, s_, dtype=np.uint8)); ps.append(np.arange(cnt, dtype=np.uint32))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt)
    return cat(rd, np.uint32), cat(fr, np.uint8), cat(ps, np.uint32)


def coherence(rd, pos, frame, ml, six: bool, n_reads: int) -> np.ndarray:
    """Compare::postProcess (Compare.hpp:2607-2728) -> float32[n_reads]; raises ReferenceThrows where the reference does."""
    sc = np.zeros(n_reads, dtype=np.float32)
    rd = np.ascontiguousarray(rd, dtype=np.uint32); pos = np.ascontiguousarray(pos, dtype=np.uint32)
    frame = np.ascontiguousarray(frame, dtype=np.uint8); ml = np.ascontiguousarray(ml, dtype=np.uint8)
    fail = C.c_uint64(0)
    rc = lib().ko_coherence(_p(rd), _p(pos), _p(frame), _p(ml), C.c_uint64(rd.shape[0]), C.c_int(int(six)), _p(sc),
                            C.c_uint64(n_reads), C.byref(fail))
    if rc != 0:
        raise ReferenceThrows(f"vector::_M_range_check: __n (which is {fail.value}) >= this->size() (which is {rd.shape[0]})")
    return sc


def identify_batch_coherence(ix, bases, offsets, p: Params, closed_form=False, lut=None):
    """identify_batch with --coherence: (CompareResult, nQueries, coherence f32[nReads], matchLen u8[nQ] in emission
    order).  Single-end input without -e (the reference's result for the other cases hangs on its unstable sorts)."""
    iv = IndexView(ix)
    km, rd = encode(bases, offsets, p, lut)
    n = int(km.shape[0])
    n_reads = offsets.shape[0] - 1
    km_s, idx_s = sort_queries(km, np.arange(n, dtype=np.uint32))   # the payload is the emission index
    rd_s = np.ascontiguousarray(rd[idx_s])
    rs, rl = ranges(iv, p, km_s)
    res, ml_s = compare_ml(iv, p, km_s, rd_s, rs, rl, n_reads, closed_form)
    ml = np.zeros(n, dtype=np.uint8)
    ml[idx_s] = ml_s
    erd, efr, eps = emission_geometry(np.asarray(offsets), p)
    assert np.array_equal(erd, rd)
    six = p.frames == 6 and not p.protein
    return res, n, coherence(erd, eps, efr, ml, six, n_reads), ml


def unique_queries(km: np.ndarray, rd: np.ndarray):
    """-e (Compare.hpp:3167-3178) on sorted records."""
    km, rd = km.copy(), rd.copy()
    n = int(lib(_K_of(km)).ko_unique_queries(_p(km), _p(rd), C.c_uint64(km.shape[0])))
    return km[:n], rd[:n]


def identify_batch(ix, bases, offsets, p: Params, want_reads=True, closed_form=False, unique=False, seg_read=None,
                   n_reads=None, lut=None):
    """Whole reference batch: encode -> sort [-> unique] -> ranges -> merge.  Returns (CompareResult, nQueries);
    nQueries counts the k-mers before -e, as iNumberOfkMersInInput does (Compare.hpp:3124)."""
    iv = IndexView(ix)
    km, rd = encode(bases, offsets, p, lut)
    n_in = int(km.shape[0])
    if seg_read is not None:   # paired-end: both mates of a pair are entries of one read (Read.hpp:834-1049)
        rd = np.ascontiguousarray(np.asarray(seg_read, dtype=np.uint32)[rd])
    km, rd = sort_queries(km, rd)
    if unique:
        km, rd = unique_queries(km, rd)
    rs, rl = ranges(iv, p, km)
    res = compare(iv, p, km, rd, rs, rl, offsets.shape[0] - 1 if n_reads is None else int(n_reads), want_reads, closed_form)
    return res, n_in


def identify_threaded(iv: IndexView, bases, offsets, p: Params, threads: int, want_tables: bool = False, lut=None):
    """The whole batch with the reference's threading model (ko_identify_threaded): for bench.py's cpu_baseline.
    -> (countAll, countUnique, nQueries) when want_tables, else nQueries."""
    L = lib(p.K)
    lut = codon_table() if lut is None else lut
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.shape[0] - 1
    nK = p.kHigh - p.kLow + 1
    ca = np.zeros((nK, iv.n_taxa), dtype=np.float64) if want_tables else None
    cu = np.zeros((nK, iv.n_taxa), dtype=np.uint64) if want_tables else None
    nq = C.c_uint64(0)
    rc = L.ko_identify_threaded(C.byref(p), C.byref(iv.c), _p(bases), _p(offsets), C.c_int64(n), _p(lut), C.c_int(int(threads)),
                                _p(ca) if want_tables else None, _p(cu) if want_tables else None, None, C.byref(nq))
    assert rc == 0
    return (ca, cu, int(nq.value)) if want_tables else int(nq.value)


def best_score(length: int, p: Params) -> np.float32:
    return np.float32(lib(p.K).ko_best_score(C.c_uint64(int(length)), C.byref(p)))


def relative_score(score, freq: int, length: int, p: Params) -> float:
    return float(lib(p.K).ko_relative_score(C.c_float(float(score)), C.c_uint64(int(freq)),
                                         C.c_uint64(int(length)), C.byref(p)))


def error_score(best, score) -> float:
    return float(lib().ko_error_score(C.c_float(float(best)), C.c_float(float(score))))
