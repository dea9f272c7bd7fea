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
