"""Seeded synthetic database + reads of SURVEY.md section 8(d): G genomes of length L, uniform ACGT, odd
genomes = 3 % mutated copy of the preceding one (shared k-mers), reads sampled uniformly with 1 %
substitutions.  The index is produced with the device's own encoder + radix sort (one "read" per
genome, read id = taxon), then made unique on the host -- the same records `build --three` would
emit for these genomes, minus its '^'-padded tail windows.
"""
from __future__ import annotations

import numpy as np

from . import capi, formats
from .reads import ReadBatch

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def genomes(n_taxa: int, length: int, seed: int, divergence: float = 0.03) -> np.ndarray:
    """u8[n_taxa, length]"""
    rng = np.random.default_rng(seed)
    g = np.empty((n_taxa, length), dtype=np.uint8)
    for t in range(n_taxa):
        if t % 2 == 1:
            g[t] = g[t - 1]
            m = rng.random(length) < divergence
            g[t, m] = _ACGT[rng.integers(0, 4, size=int(m.sum()))]
        else:
            g[t] = _ACGT[rng.integers(0, 4, size=length)]
    return g


def genomes_crowded(n_taxa: int, length: int, seed: int, clade_share: float = 0.30, universal_share: float = 0.05,
                    clade_sizes=(50, 200), divergence=(0.01, 0.05), gene_len=(1000, 3000)) -> np.ndarray:
    """u8[n_taxa, length]: taxa in clades of 50-200 that share conserved "genes" -- what real bacterial indices look like at
    k = 7...9 (hard part 5; Compare.hpp:396-441,917-955 is where the reference pays for it).  A clade
    has an ancestor and a set of gene intervals covering `clade_share` of the genome: every member copies those intervals
    from the ancestor with its own 1-5 % of substitutions; `universal_share` of every genome comes the same way from ONE
    universal ancestor; the rest is the taxon's own random sequence.  So a conserved k-mer at k = 7 sits in tens to
    hundreds of taxa and at k = 12 in a handful."""
    rng = np.random.default_rng(seed)

    def intervals(share):
        mask = np.zeros(length, dtype=bool)
        want, guard = int(share * length), 0
        while mask.sum() < want and guard < 100000:
            n = int(rng.integers(gene_len[0], gene_len[1] + 1))
            a = int(rng.integers(0, max(1, length - n)))
            mask[a:a + n] = True
            guard += 1
        return mask

    def mutated(src, mask, d):
        out = src[mask].copy()
        m = rng.random(out.shape[0]) < d
        out[m] = _ACGT[rng.integers(0, 4, size=int(m.sum()))]
        return out

    g = np.empty((n_taxa, length), dtype=np.uint8)
    universal = _ACGT[rng.integers(0, 4, size=length)]
    umask = intervals(universal_share)
    t = 0
    while t < n_taxa:
        size = min(n_taxa - t, int(rng.integers(clade_sizes[0], clade_sizes[1] + 1)))
        ancestor = _ACGT[rng.integers(0, 4, size=length)]
        cmask = intervals(clade_share) & ~umask
        for x in range(t, t + size):
            g[x] = _ACGT[rng.integers(0, 4, size=length)]
            g[x, cmask] = mutated(ancestor, cmask, rng.uniform(*divergence))
            g[x, umask] = mutated(universal, umask, rng.uniform(*divergence))
        t += size
    return g


def content_for(n_taxa: int) -> formats.Content:
    return formats.Content(["non_unique"] + [f"Taxon {t}" for t in range(n_taxa)],
                           np.concatenate(([0], 100 + np.arange(n_taxa))).astype(np.uint32))


def index_from_genomes(g: np.ndarray, device: int = 0, K: int = formats.K64) -> formats.Index:
    """Encode every genome on the device (3 frames, K = 12 or 25 letters), sort, unique -> formats.Index."""
    n_taxa, length = g.shape
    content = content_for(n_taxa)
    one = np.array([1], dtype=np.uint64)
    if K > formats.K64:
        one = np.zeros(1, dtype=formats.KEY128_DTYPE)
        one["lo"] = 1
    boot = formats.make_index(one, np.array([100], dtype=np.uint32), content)
    dix = capi.DeviceIndex(boot, device, check_trie=False)
    ctx = capi.Context(dix, K, K, 3)             # kLow = K: no X marker, every window is a full K-mer
    off = np.arange(n_taxa + 1, dtype=np.int64) * length
    ctx.upload(g.reshape(-1), off)
    ctx.encode()
    ctx.sort_and_range()
    km, rd = ctx.queries()                        # sorted by k-mer, ties in genome order (stable)
    ctx.close()
    dix.close()
    tid = content.taxids[rd + 1]
    keep = np.ones(km.shape[0], dtype=bool)
    keep[1:] = (km[1:] != km[:-1]) | (tid[1:] != tid[:-1])
    km, tid = km[keep], tid[keep]
    tax = (rd[keep] + 1).astype(np.uint32)
    tp, tc = formats.trie_from_kmers(km)
    freq = np.zeros((content.n_taxa, K), dtype=np.uint64)
    cnt = np.bincount(tax, minlength=content.n_taxa).astype(np.uint64)
    freq[:] = cnt[:, None]                        # no '^' letters in these k-mers: same count at every k
    return formats.Index(km, tid.astype(np.uint32), tax, tp, tc, content, freq)


def reads_from_genomes(g: np.ndarray, n_reads: int, read_len: int, seed: int, sub_rate: float = 0.01,
                       chunk: int = 1 << 20) -> ReadBatch:
    n_taxa, length = g.shape
    flat = g.reshape(-1)
    rng = np.random.default_rng(seed)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int64)[None, :]
    for a in range(0, n_reads, chunk):
        b = min(n_reads, a + chunk)
        t = rng.integers(0, n_taxa, size=b - a)
        pos = rng.integers(0, length - read_len + 1, size=b - a)
        start = t.astype(np.int64) * length + pos
        blk = flat[start[:, None] + ar]
        m = rng.random(blk.shape) < sub_rate
        blk[m] = _ACGT[rng.integers(0, 4, size=int(m.sum()))]
        out[a:b] = blk
    off = np.arange(n_reads + 1, dtype=np.int64) * read_len
    return ReadBatch(out.reshape(-1), off, None, np.full(n_reads, read_len + 1, dtype=np.uint32))
