"""On-disk artefacts of a kASA index, read and written as flat numpy arrays.

The reference keeps these files behind STXXL vectors; they are plain packed arrays:

* ``<idx>``          packed ``{u64 kmer, u32 taxid}`` records, 12 B each, sorted by (kmer, taxid)
                     (reference: source/utils/packedPairs.hpp:107-130, source/MetaHeader.h:137).
                     The file is zero-padded to a block multiple -- trust ``_info.txt``.
* ``<idx>_info.txt`` record count [+ ``128`` / ``3`` marker] (source/modes/Build.hpp:466-470,
                     source/modes/Compare.hpp:96-109).
* ``<idx>_trie``     packed ``{u64 count, u32 prefix30}`` 12 B (source/utils/packedPairs.hpp:157-167,
                     source/modes/Trie.hpp:365-394); ``<idx>_trie.txt`` holds the entry count.
* ``<idx>_f.txt``    ``name \\t f(k=K) \\t f(K-1) ... f(1)`` (source/kASA.hpp:549-570).
* content file       ``name \\t taxid \\t taxids \\t accessions [\\t intId]``
                     (source/modes/Compare.hpp:111-151).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

REC_DTYPE = np.dtype([("kmer", "<u8"), ("tax", "<u4")], align=False)  # 12 B
TRIE_DTYPE = np.dtype([("count", "<u8"), ("prefix", "<u4")], align=False)  # 12 B
HALF_DTYPE = np.dtype([("low", "<u4"), ("tax", "<u2")], align=False)  # 6 B (shrink strategy 2)

REC128_DTYPE = np.dtype([("lo", "<u8"), ("hi", "<u8"), ("tax", "<u4")], align=False)  # 20 B (packedLargePair)
KEY128_DTYPE = np.dtype([("lo", "<u8"), ("hi", "<u8")], align=False)  # a 128-bit k-mer: low word first

K64 = 12  # letters per packed k-mer in a 64-bit index
K128 = 25  # ... in a 128-bit index (source/utils/uint128_t.hpp, main.cpp --kH)
TRIE_LETTERS = 6


def is_wide(kmer: np.ndarray) -> bool:
    return kmer.dtype == KEY128_DTYPE


def key_shr(kmer: np.ndarray, s: int) -> np.ndarray:
    """Low 64 bits of (kmer >> s) for u64 keys and for (lo, hi) 128-bit keys."""
    if not is_wide(kmer):
        return kmer >> np.uint64(s) if s < 64 else np.zeros(kmer.shape, np.uint64)
    lo, hi = kmer["lo"], kmer["hi"]
    if s == 0:
        return lo.copy()
    if s >= 64:
        return hi >> np.uint64(s - 64) if s < 128 else np.zeros(lo.shape, np.uint64)
    return (lo >> np.uint64(s)) | (hi << np.uint64(64 - s))


def key_order(kmer: np.ndarray, *minor) -> np.ndarray:
    """argsort by numeric key value (then by the minor keys, first = least significant... as np.lexsort)."""
    if is_wide(kmer):
        return np.lexsort(tuple(minor) + (kmer["lo"], kmer["hi"]))
    return np.lexsort(tuple(minor) + (kmer,))


@dataclass
class Content:
    names: list          # names[0] == "non_unique"
    taxids: np.ndarray   # u32[nTaxa]; taxids[0] == 0

    @property
    def n_taxa(self) -> int:
        return len(self.names)


@dataclass
class Index:
    """A kASA index in host memory (dense taxon indices already applied).  64-bit index: u64 keys of 12 letters;
    128-bit index: KEY128_DTYPE keys of 25 letters."""
    kmer: np.ndarray        # u64[n] or KEY128_DTYPE[n], ascending
    taxid: np.ndarray       # u32[n], original tax IDs from the file
    tax: np.ndarray         # u32[n], dense taxon index via the content file
    trie_prefix: np.ndarray  # u32[m], ascending 30-bit prefixes
    trie_count: np.ndarray   # u64[m]
    content: Content
    freq: np.ndarray        # u64[nTaxa, K]: freq[t, j] = k-mers of taxon t at k = K - j

    @property
    def K(self) -> int:
        return K128 if is_wide(self.kmer) else K64

    @property
    def n(self) -> int:
        return int(self.kmer.shape[0])

    @property
    def trie_start(self) -> np.ndarray:
        s = np.zeros(self.trie_count.shape[0], dtype=np.uint64)
        if s.shape[0] > 1:
            np.cumsum(self.trie_count[:-1], out=s[1:])
        return s

    @property
    def trie_len_m1(self) -> np.ndarray:
        return (self.trie_count - np.uint64(1)).astype(np.uint32)

    def freq_at(self, k: int) -> np.ndarray:
        """k-mer count of every taxon at length k (what Compare.hpp:166-179 loads per level)."""
        return self.freq[:, self.K - k]


def read_info(prefix: str):
    with open(prefix + "_info.txt") as f:
        tok = f.read().split()
    n = int(tok[0])
    kind = int(tok[1]) if len(tok) > 1 else 0
    return n, kind


def read_records(prefix: str):
    n, kind = read_info(prefix)
    if kind == 128:
        rec = np.fromfile(prefix, dtype=REC128_DTYPE, count=n)
        if rec.shape[0] != n:
            raise RuntimeError("The index file is shorter than _info.txt says")
        km = np.zeros(n, dtype=KEY128_DTYPE)
        km["lo"], km["hi"] = rec["lo"], rec["hi"]
        return km, np.ascontiguousarray(rec["tax"])
    if kind == 3:
        return None, None  # halved records: rebuilt by load_index with the trie and the content file
    rec = np.fromfile(prefix, dtype=REC_DTYPE, count=n)
    if rec.shape[0] != n:
        raise RuntimeError("The index file is shorter than _info.txt says")
    return np.ascontiguousarray(rec["kmer"]), np.ascontiguousarray(rec["tax"])


def read_trie(prefix: str):
    if not (os.path.exists(prefix + "_trie.txt") and os.path.exists(prefix + "_trie")):
        raise RuntimeError("The trie file cannot be found!")  # Compare.hpp:331-333
    with open(prefix + "_trie.txt") as f:
        m = int(f.read().split()[0])
    t = np.fromfile(prefix + "_trie", dtype=TRIE_DTYPE, count=m)
    return np.ascontiguousarray(t["prefix"]), np.ascontiguousarray(t["count"])


def read_content(path: str) -> Content:
    names, taxids = ["non_unique"], [0]
    as_str = False
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line == "":
                continue
            cols = line.split("\t")
            if len(cols) >= 5:
                as_str = True
            if len(cols) < 4:
                raise RuntimeError(
                    "Content file contains less than 4 columns, it may be damaged... "
                    "The faulty line was: " + line + "\n")
            names.append(cols[0].replace(",", ""))
            taxids.append(int(cols[4]) if as_str else int(cols[1]))
    return Content(names, np.asarray(taxids, dtype=np.uint32))


def read_freq(prefix: str, n_taxa: int, K: int = K64) -> np.ndarray:
    out = np.zeros((n_taxa, K), dtype=np.uint64)
    row = 0
    with open(prefix + "_f.txt") as f:
        for line in f:
            line = line.rstrip("\n")
            if line == "":
                continue
            cols = line.split("\t")
            vals = [int(x) for x in cols[1:]]
            if row < n_taxa:
                out[row, :min(len(vals), K)] = vals[:K]
            row += 1
    return out


def dense_tax(taxid: np.ndarray, content: Content) -> np.ndarray:
    """taxid -> dense index 1..nTaxa-1 (Compare.hpp:139-143's mTaxToIdx)."""
    order = np.argsort(content.taxids, kind="stable")
    sorted_ids = content.taxids[order]
    pos = np.searchsorted(sorted_ids, taxid)
    pos = np.minimum(pos, sorted_ids.shape[0] - 1)
    if not np.array_equal(sorted_ids[pos], taxid):
        raise RuntimeError("index holds a tax ID the content file does not know")
    return order[pos].astype(np.uint32)


def load_index(prefix: str, content_path: str) -> Index:
    if not os.path.exists(prefix):
        raise RuntimeError("The index file cannot be found!")  # Compare.hpp:97-99
    content = read_content(content_path)
    kmer, taxid = read_records(prefix)
    tp, tc = read_trie(prefix)
    if kmer is None:
        # shrink strategy 2 (source/modes/Shrink.hpp:78-143): {u32 low 30 bits, u16 dense taxon index} per entry,
        # the upper 30 bits of an entry are its _trie prefix
        n, _ = read_info(prefix)
        half = np.fromfile(prefix, dtype=HALF_DTYPE, count=n)
        if int(tc.sum()) != n:
            raise RuntimeError("The trie file does not match the halved index")
        pre = np.repeat(tp.astype(np.uint64), tc.astype(np.int64))
        kmer = (pre << np.uint64(30)) | (half["low"].astype(np.uint64) & np.uint64(0x3FFFFFFF))
        taxid = content.taxids[half["tax"].astype(np.int64)]
    freq = read_freq(prefix, content.n_taxa, K128 if is_wide(kmer) else K64)
    return Index(kmer, taxid, dense_tax(taxid, content), tp, tc, content, freq)


# ------------------------------------------------------------------------------------------------
# Writers (synthetic indices for tests and bench; same layout as the reference's build mode emits)
# ------------------------------------------------------------------------------------------------

def trie_from_kmers(kmer: np.ndarray):
    """prefix30 -> count table of a sorted k-mer array (what Trie.hpp:365-394 writes)."""
    K = K128 if is_wide(kmer) else K64
    pre = key_shr(kmer, 5 * (K - TRIE_LETTERS)).astype(np.uint32)
    if pre.shape[0] == 0:
        return pre, np.zeros(0, dtype=np.uint64)
    change = np.flatnonzero(np.concatenate(([True], pre[1:] != pre[:-1])))
    counts = np.diff(np.concatenate((change, [pre.shape[0]]))).astype(np.uint64)
    return pre[change], counts


def freq_from_index(kmer: np.ndarray, tax: np.ndarray, n_taxa: int) -> np.ndarray:
    """kASA.hpp:517-526: for j = 0..K-1 count entries whose letter (from the right) j is not '^'."""
    K = K128 if is_wide(kmer) else K64
    out = np.zeros((n_taxa, K), dtype=np.uint64)
    for j in range(K):
        ok = (key_shr(kmer, 5 * j) & np.uint64(31)) != np.uint64(30)
        out[:, j] = np.bincount(tax[ok], minlength=n_taxa).astype(np.uint64)
    return out


def make_index(kmer: np.ndarray, taxid: np.ndarray, content: Content) -> Index:
    """Sort + unique (kmer, taxid) pairs into an Index (Build.hpp:305-356's net effect)."""
    order = key_order(kmer, taxid)
    kmer, taxid = kmer[order], taxid[order]
    if kmer.shape[0]:
        keep = np.concatenate(([True], (kmer[1:] != kmer[:-1]) | (taxid[1:] != taxid[:-1])))
        kmer, taxid = kmer[keep], taxid[keep]
    tax = dense_tax(taxid, content)
    tp, tc = trie_from_kmers(kmer)
    return Index(np.ascontiguousarray(kmer), np.ascontiguousarray(taxid.astype(np.uint32)), tax, tp, tc,
                 content, freq_from_index(kmer, tax, content.n_taxa))


def write_index(ix: Index, prefix: str, content_path: str) -> None:
    if is_wide(ix.kmer):
        rec = np.zeros(ix.n, dtype=REC128_DTYPE)
        rec["lo"], rec["hi"], rec["tax"] = ix.kmer["lo"], ix.kmer["hi"], ix.taxid
    else:
        rec = np.zeros(ix.n, dtype=REC_DTYPE)
        rec["kmer"], rec["tax"] = ix.kmer, ix.taxid
    rec.tofile(prefix)
    with open(prefix + "_info.txt", "w") as f:
        f.write(str(ix.n) + ("\n128" if is_wide(ix.kmer) else ""))
    t = np.zeros(ix.trie_prefix.shape[0], dtype=TRIE_DTYPE)
    t["count"], t["prefix"] = ix.trie_count, ix.trie_prefix
    t.tofile(prefix + "_trie")
    with open(prefix + "_trie.txt", "w") as f:
        f.write(str(t.shape[0]))
    with open(prefix + "_f.txt", "w") as f:
        for r in range(ix.content.n_taxa):
            f.write(ix.content.names[r] + "\t" + "\t".join(str(int(v)) for v in ix.freq[r]) + "\n")
    with open(content_path, "w") as f:
        for r in range(1, ix.content.n_taxa):
            tid = int(ix.content.taxids[r])
            f.write(f"{ix.content.names[r]}\t{tid}\t{tid}\tACC{r}\n")


def write_index_halved(ix: Index, prefix: str) -> None:
    """The index of `shrink -s 2` (source/modes/Shrink.hpp:78-143): entries with fewer than seven real letters (their
    seventh letter is already '^') are dropped, the others keep the low 30 bits of the k-mer and the dense taxon index
    -- {u32, u16} = 6 bytes -- next to a `_trie` that supplies the upper 30 bits.  64-bit indices only."""
    if is_wide(ix.kmer):
        raise NotImplementedError("the halved format exists for 64-bit indices only")
    keep = ((ix.kmer >> np.uint64(25)) & np.uint64(31)) != np.uint64(30)
    km, tax, taxid = ix.kmer[keep], ix.tax[keep], ix.taxid[keep]
    rec = np.zeros(km.shape[0], dtype=HALF_DTYPE)
    rec["low"] = (km & np.uint64(0x3FFFFFFF)).astype(np.uint32)
    rec["tax"] = tax.astype(np.uint16)
    rec.tofile(prefix)
    with open(prefix + "_info.txt", "w") as f:
        f.write(str(km.shape[0]) + "\n3")
    tp, tc = trie_from_kmers(km)
    t = np.zeros(tp.shape[0], dtype=TRIE_DTYPE)
    t["count"], t["prefix"] = tc, tp
    t.tofile(prefix + "_trie")
    with open(prefix + "_trie.txt", "w") as f:
        f.write(str(t.shape[0]))
    with open(prefix + "_f.txt", "w") as f:                          # shrink copies the frequency file of the full index
        for r in range(ix.content.n_taxa):
            f.write(ix.content.names[r] + "\t" + "\t".join(str(int(v)) for v in ix.freq[r]) + "\n")
