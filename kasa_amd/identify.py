"""Host driver of `identify`: the per-batch call sequence of Compare::CompareWithLib_partialSort
(source/modes/Compare.hpp:2733-3766) on top of the C ABI.

    read batch (Read.hpp:1054) -> kasa_batch_upload/encode -> kasa_batch_sort_and_range
    -> kasa_batch_lookup_score -> rank + write per-read text (Compare.hpp:1485-1890)
    after the file: profile CSV (Compare.hpp:3466-3665)

Batch boundaries matter for the last float digit of per-read scores (the flush order depends on
which reads share a batch).  `memory_gib` (the reference's -m) cuts the batches exactly
where `kASA identify -m <GiB>` cuts them (capi.RefBatcher, kasa_refbatch_* of the C ABI), which makes the per-read file
byte-identical to the reference's for any input size; without it (`batch_reads=None`) the whole input is one batch,
which is what the reference does whenever the input fits its -m budget.
"""
from __future__ import annotations

import numpy as np

from . import capi, report
from .formats import Index
from .reads import ReadBatch


class Identify:
    def __init__(self, index: Index, device: int = 0, k_high: int = 12, k_low: int = 7, frames: int = 3,
                 threshold: float = 0.0, beasts: int = 3, fmt: str = "json", dix: capi.DeviceIndex = None,
                 unique: bool = False, codon_lut=None, coherence: bool = False, coherence_threshold: float = 11.0):
        self.index = index
        self.k_high, self.k_low = max(k_high, k_low), min(k_high, k_low)
        self.frames, self.threshold, self.beasts, self.fmt = frames, threshold, beasts, fmt
        self.dix = dix if dix is not None else capi.DeviceIndex(index, device)
        self.ctx = capi.Context(self.dix, self.k_high, self.k_low, frames, codon_lut)
        self.unique = unique
        self.coherence = coherence      # --coherence (Compare::postProcess): one more number per read, and a second --filter rule
        self.coherence_threshold = np.float32(coherence_threshold)
        self.contaminants = []          # read numbers --filter would move to the contaminants (report.is_contaminant)
        self.error_threshold = 0.5
        self.n_kmers = 0
        self.n_reads = 0
        self.device_rank = True         # kasa_batch_rank; False: the whole CSR comes back and the host ranks every read
        self.device_text = True         # kasa_batch_text: the device writes the per-read file's bytes (needs device_rank, no flagged reads)
        self.device_text_batches = 0    # batches whose text came from the device
        self.flagged_reads = 0          # reads the device handed back to the host's std::sort emulation
        self.piece_bytes = None         # tests: another limit than the reference's 100 MiB for the pieces of a long sequence (reads.py)
        self.piece_bounds = None        # tests: the batches over the pieces, [0, p1, ..., nPieces], instead of the reference's (-m) or one batch

    def close(self):
        self.ctx.close()

    def run(self, reads: ReadBatch, want_per_read: bool = True, batch_reads: int = None, coverage: bool = False,
            memory_gib: int = None, threads: int = 1, ram: bool = False, keep_csr: bool = False):
        """-> (per-read text or None, profile CSV text, list of CSR batches (with keep_csr))."""
        ix = self.index
        pieced = reads if not reads.layout else (reads.with_pieces(ix.K, self.frames, self.coherence) if self.piece_bytes is None else
                                                 reads.with_pieces(ix.K, self.frames, self.coherence, self.piece_bytes))
        if pieced is not reads:
            return self._run_pieced(pieced, want_per_read, coverage, memory_gib, threads, ram, keep_csr)
        writer = report.ReadWriter(self.fmt, ix.content.names, ix.content.taxids, self.beasts, coherence=self.coherence)
        freq = ix.freq_at(self.k_high)
        out = [writer.header()] if want_per_read else None
        csr = []
        self.ctx.profile_reset()
        protein = bool(reads.protein)
        self.ctx.set_protein(protein)
        self.n_kmers = 0
        self.n_reads = 0
        self.flagged_reads = 0
        self.contaminants = []
        self.device_text_batches = 0
        if want_per_read and self.device_text and self.device_rank:
            self.ctx.set_taxa_text(ix.content.taxids, ix.content.names)
        step = reads.n if not batch_reads else batch_reads
        if not batch_reads and reads.n:
            # the whole input is one batch unless it does not fit the free HBM (the reference cuts batches by its -m budget)
            per_read = (int(reads.offsets[-1] - reads.offsets[0]) // reads.n + 64) * (2 if self.frames == 6 else 1)
            fit = self.ctx.max_queries_per_batch(self.ctx.dix.device if hasattr(self.ctx.dix, "device") else 0) // max(1, per_read)
            step = max(1, min(reads.n, fit))
        bounds = None
        if memory_gib is not None and want_per_read and reads.n:
            # the reference's own batch boundaries (-m): the per-read float sums depend on them
            bounds = capi.RefBatcher(ix, self.k_high, self.k_low, self.frames, memory_gib, threads, ram,
                                     record_bytes=getattr(ix, "record_bytes", None), coherence=self.coherence).boundaries(reads, True)
        self.batch_sizes = []
        a = 0
        while a < reads.n or (a == 0 and reads.n == 0):
            b = min(reads.n, a + max(step, 1)) if bounds is None else bounds[len(self.batch_sizes) + 1]
            self.batch_sizes.append(b - a)
            part = reads.slice(a, b)
            self.ctx.run_batch(part.bases, part.offsets, want_per_read, coverage, self.unique, part.seg_read, part.n)
            self.n_kmers += self.ctx.n_kmers
            coh = self.ctx.coherence() if (want_per_read and self.coherence) else None   # Compare.hpp:3317-3321
            if want_per_read:
                # ranked on the device (kasa_batch_rank): only what the writer can print crosses PCIe; the full CSR comes
                # back for the reads the device flags (ties under an unstable sort) or on request (keep_csr)
                meta = ent = None
                device_rank = self.device_rank and part.n > 0 and np.unique(part.lengths).shape[0] * len(freq) <= 4_000_000
                flagged = 0
                if device_rank:
                    den, rclass = report.rank_denominators(freq, part.lengths, ix.K, protein)
                    meta, ent, flagged = self.ctx.rank(den, rclass, float(np.float32(self.threshold)), self.beasts)
                self.flagged_reads += flagged
                off = tax = sc = None
                if keep_csr or not device_rank or flagged:
                    off, tax, sc = self.ctx.scores(pinned=not keep_csr)
                    if keep_csr:
                        csr.append((off, tax, sc))
                if device_rank and self.device_text and not flagged and not keep_csr:
                    # the text is written on the device (kasa_batch_text): the hits do not cross PCIe at all
                    distinct = np.unique(np.asarray(part.lengths))
                    best = np.array([report.best_score(int(L), self.k_high, self.k_low, self.frames, protein) for L in distinct], dtype=np.float32)
                    text, _, cont = self.ctx.text(self.fmt, self.beasts, self.n_reads, part.names, part.lengths, best, coherence=coh is not None,
                                                  error_threshold=self.error_threshold, coherence_threshold=self.coherence_threshold,
                                                  pieces=1 + self.device_text_batches % 3)
                    out.append(text.decode("latin-1"))
                    self.contaminants.extend(int(self.n_reads + r) for r in np.flatnonzero(cont))
                    self.device_text_batches += 1
                    self.n_reads += part.n
                    a = b
                    if reads.n == 0:
                        break
                    continue
                for r in range(part.n):
                    length = int(part.lengths[r])
                    if device_rank and not (int(meta[r, 1]) >> 31):
                        a0, n0 = int(meta[r, 0]), int(meta[r, 1]) & 0x7FFFFFFF
                        max_score = np.uint32(meta[r, 2]).view(np.float32)
                        rk = report.ranked_from_prefix(ent[a0:a0 + n0], max_score, length, self.k_high, self.k_low, self.frames,
                                                       self.beasts, protein)
                    else:
                        lo, hi = int(off[r]), int(off[r + 1])
                        rk = report.rank_read(tax[lo:hi], sc[lo:hi], length, freq, self.k_high,
                                              self.k_low, self.frames, self.threshold, self.beasts, K=ix.K, protein=protein)
                        max_score = max((h.score for h in rk.hits), default=np.float32(0))
                    out.append(writer.read(self.n_reads + r, part.names[r], length, rk, None if coh is None else coh[r]))
                    if rk.hits and (report.is_contaminant(rk.best, max_score, self.error_threshold) or
                                    (coh is not None and coh[r] >= self.coherence_threshold)):   # Compare.hpp:1597-1606
                        self.contaminants.append(self.n_reads + r)
            self.n_reads += part.n
            a = b
            if reads.n == 0:
                break
        if want_per_read:
            out.append(writer.footer())
        ca, cu, ct = self.ctx.profile()
        freq_lv = np.stack([ix.freq_at(k) for k in range(self.k_high, self.k_low - 1, -1)], axis=1) if coverage else None
        prof = report.profile_csv(ca, cu, ix.content.names, ix.content.taxids, self.k_high, self.k_low,
                                  self.n_kmers, self.n_reads, 3 if (protein and self.frames == 6) else self.frames,
                                  count_total=ct if coverage else None, freq=freq_lv)
        return ("".join(out) if want_per_read else None), prof, csr

    def _run_pieced(self, pieced: ReadBatch, want_per_read: bool, coverage: bool, memory_gib: int, threads: int, ram: bool, keep_csr: bool):
        """An input with sequences the reference reads in PIECES (reads.py; Read.hpp:371-600): the pieces of a read are
        sequences of one read id on the device, and where a batch ends between two of them (Read.hpp:1147-1186, strTransfer)
        the read's scores so far wait on the host for the rest (Compare::saveResults, Compare.hpp:2324-2443): the taxa found
        on both sides get the float sum, and the read is ranked and printed when its last piece has been scored.  Ranking is
        the host's here (a file of contigs has few reads)."""
        if self.coherence:
            raise RuntimeError("--coherence over sequences long enough for kASA to read them in pieces is not supported")
        ix = self.index
        writer = report.ReadWriter(self.fmt, ix.content.names, ix.content.taxids, self.beasts)
        freq = ix.freq_at(self.k_high)
        out = [writer.header()] if want_per_read else None
        self.ctx.profile_reset()
        protein = bool(pieced.protein)
        self.ctx.set_protein(protein)
        self.n_kmers = self.n_reads = self.flagged_reads = self.device_text_batches = 0
        self.contaminants = []
        seg = pieced.seg_read.astype(np.int64)
        n_pieces = len(seg)
        if self.piece_bounds is not None:
            bounds = list(self.piece_bounds)
        elif memory_gib is not None:
            bounds = capi.RefBatcher(ix, self.k_high, self.k_low, self.frames, memory_gib, threads, ram, record_bytes=getattr(ix, "record_bytes", None),
                                     coherence=False).piece_batches(pieced, want_per_read)
        else:
            bounds = [0, n_pieces]
        self.batch_sizes = []
        saved_tax, saved_sc = np.zeros(0, np.uint32), np.zeros(0, np.float32)
        carried = 0            # strTransfer::lengthOfDNA (Read.hpp:1181: it grows by the running length at every unfinished piece)
        csr = []

        def merge(t0, s0, t1, s1):
            allt = np.union1d(t0, t1)
            acc = np.zeros(allt.shape[0], np.float32)
            has = np.zeros(allt.shape[0], bool)
            for t, s in ((t0, s0), (t1, s1)):
                at = np.searchsorted(allt, t)
                acc[at] = np.where(has[at], (acc[at] + s).astype(np.float32), s)
                has[at] = True
            return allt.astype(np.uint32), acc

        for pa, pb in zip(bounds[:-1], bounds[1:]):
            r0 = int(seg[pa])
            local = (seg[pa:pb] - r0).astype(np.uint32)
            n_local = int(local[-1]) + 1
            tail = pb < n_pieces and seg[pb] == seg[pb - 1]          # the last read goes on in the next batch (addTail)
            o = pieced.offsets[pa:pb + 1]
            self.batch_sizes.append(n_local)
            self.ctx.run_batch(pieced.bases[int(o[0]):int(o[-1])], o - o[0], want_per_read, coverage, self.unique, local, n_local)
            self.n_kmers += self.ctx.n_kmers
            # "Length" as the reader counts it across its calls (Read.hpp:1117,1166-1186)
            length = carried
            lengths = np.zeros(n_local, np.int64)
            for q in range(pa, pb):
                length += int(pieced.piece_chars[q])
                lr = int(local[q - pa])
                is_last = q + 1 == n_pieces or seg[q + 1] != seg[q]
                if is_last:
                    lengths[lr], length, carried = length, 0, 0
                else:
                    carried += length
            if want_per_read:
                off, tax, sc = self.ctx.scores(pinned=not keep_csr)
                if keep_csr:
                    csr.append((off, tax, sc))
                # Compare::saveResults: what is waiting joins read 0 only when the batch ends with a finished read (:2344);
                # otherwise it stays and takes the unfinished read's row as well (:2388-2409) -- the reference's own rule
                for lr in range(n_local):
                    lo, hi = int(off[lr]), int(off[lr + 1])
                    t, s_ = tax[lo:hi], sc[lo:hi]
                    if lr == 0 and saved_tax.shape[0] and not tail:
                        t, s_ = merge(saved_tax, saved_sc, t, s_)
                        saved_tax, saved_sc = np.zeros(0, np.uint32), np.zeros(0, np.float32)
                    if lr == n_local - 1 and tail:
                        if hi > lo:
                            saved_tax, saved_sc = merge(saved_tax, saved_sc, t, s_)
                        continue
                    rd = r0 + lr
                    ln = int(lengths[lr]) & 0xFFFFFFFF
                    rk = report.rank_read(t, s_, ln, freq, self.k_high, self.k_low, self.frames, self.threshold, self.beasts, K=ix.K, protein=protein)
                    max_score = max((h.score for h in rk.hits), default=np.float32(0))
                    out.append(writer.read(rd, pieced.names[rd], ln, rk, None))
                    if rk.hits and report.is_contaminant(rk.best, max_score, self.error_threshold):
                        self.contaminants.append(rd)
            self.n_reads = r0 + n_local - (1 if tail else 0)
        if want_per_read:
            out.append(writer.footer())
        ca, cu, ct = self.ctx.profile()
        freq_lv = np.stack([ix.freq_at(k) for k in range(self.k_high, self.k_low - 1, -1)], axis=1) if coverage else None
        prof = report.profile_csv(ca, cu, ix.content.names, ix.content.taxids, self.k_high, self.k_low, self.n_kmers, self.n_reads,
                                  3 if (protein and self.frames == 6) else self.frames, count_total=ct if coverage else None, freq=freq_lv)
        return ("".join(out) if want_per_read else None), prof, csr
