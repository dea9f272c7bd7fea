"""Shared test helpers: run the ORACLE over a whole input the way the reference runs a batch, and
render it with the product's host-side report module."""
import os

import numpy as np

from kasa_amd import formats, reads, report
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name, stem="idx"):
    d = os.path.join(GOLDEN, name)
    return d, formats.load_index(os.path.join(d, stem), os.path.join(d, "content.txt"))


def csr_from_dense(M):
    """Non-zero cells per read, taxon index ascending (what Compare.hpp:1501-1522 scans)."""
    rows = []
    for r in range(M.shape[0]):
        t = np.flatnonzero(M[r, 1:] > 0) + 1
        rows.append((t.astype(np.uint32), M[r, t].astype(np.float32)))
    return rows


def render(ix, batch, rows, count_all, count_unique, n_kmers, fmt, k_high, k_low, frames,
           threshold=0.0, beasts=3, protein=False, count_total=None, coherence=None):
    w = report.ReadWriter(fmt, ix.content.names, ix.content.taxids, beasts, coherence=coherence is not None)
    freq = ix.freq_at(k_high)
    out = [w.header()]
    for r in range(batch.n):
        t, s = rows[r]
        rk = report.rank_read(t, s, int(batch.lengths[r]), freq, k_high, k_low, frames, threshold, beasts,
                              K=ix.K, protein=protein)
        out.append(w.read(r, batch.names[r], int(batch.lengths[r]), rk, None if coherence is None else coherence[r]))
    out.append(w.footer())
    prof = report.profile_csv(count_all, count_unique, ix.content.names, ix.content.taxids, k_high, k_low,
                              n_kmers, batch.n, 3 if (protein and frames == 6) else frames,
                              count_total=count_total,
                              freq=None if count_total is None else
                              np.stack([ix.freq_at(k) for k in range(k_high, k_low - 1, -1)], axis=1))
    return "".join(out), prof


def oracle_identify(ix, batch, k_high=12, k_low=7, frames=3, avx_quirk=False, closed_form=False, unique=False,
                    protein=False, cmp64_quirk=False, coverage=False, lut=None):
    p = oracle.params(k_high, k_low, frames, avx_quirk, coverage=coverage, K=ix.K, protein=protein,
                      cmp64_quirk=cmp64_quirk)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True, closed_form, unique,
                                    seg_read=batch.seg_read, n_reads=batch.n, lut=lut)
    return res, nq
