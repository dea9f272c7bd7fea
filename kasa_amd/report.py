"""Per-read ranking + text output and the profile table, as the reference writes them.

Host-side half of the `identify` path: consumes the CSR of non-zero
(read, taxon) scores the device returns and the profile tables, produces the same bytes as

* source/modes/Compare.hpp:1452-1481  (best score),
* source/modes/Compare.hpp:1485-1890  (ranking, top/further hits, JSON / JSONL / TSV / Kraken text),
* source/modes/Compare.hpp:3466-3665  (profile CSV).

Arithmetic follows the reference's types: k-mer scores and the error are float32, the relative score
is float64 with libm's log2 (math.log2 is the same libm call).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .textnum import dtoa, itoa

K64 = 12
_F32 = np.float32


def weight(k: int) -> np.float32:
    """Compare.hpp:392: w_k = k^2 / 625 in float."""
    return _F32(k * k) / _F32(625.0)


def best_score(length: int, k_high: int, k_low: int, frames: int, protein: bool = False) -> np.float32:
    """Compare.hpp:1452-1481.  The subtraction is unsigned 64-bit in the reference."""
    best = _F32(0.0)
    for i in range(k_low, k_high + 1):
        if protein:
            span = (length - i + 1) & 0xFFFFFFFFFFFFFFFF
        elif frames == 1:
            span = (length // 3 - i + 1) & 0xFFFFFFFFFFFFFFFF
        elif frames == 6:
            span = (2 * ((length - 3 * i + 1) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
        else:
            span = (length - 3 * i + 1) & 0xFFFFFFFFFFFFFFFF
        best = _F32(best + _F32(_F32(span) * weight(i)))
    return best


def relative_score(score: np.float32, freq: int, length: int, K: int = K64, protein: bool = False) -> float:
    """Compare.hpp:1506-1511: score / (1 + log2(freq * double(len - 3K + 1))) (len - K + 1 for protein
    input), uint32 subtraction."""
    span = (int(length) - (K if protein else 3 * K) + 1) & 0xFFFFFFFF
    prod = float(int(freq)) * float(span)
    if prod <= 0.0:
        lg = -math.inf if prod == 0.0 else math.nan
    else:
        lg = math.log2(prod)
    den = 1.0 + lg
    s = float(score)
    if den == 0.0:
        return math.inf if s > 0 else math.nan
    return s / den


@dataclass
class Hit:
    tax_idx: int
    score: np.float32
    rel: float


@dataclass
class Ranked:
    hits: list        # sorted by relative score, descending
    n_top: int
    best: np.float32


def _stdsort_order(n: int, less):
    """Ids 0..n-1 in the order libstdc++'s std::sort leaves them (less(x, y): x goes before y): introsort with the median
    of three moved to the front, unguarded partition, ranges of at most 16 left to one final insertion sort -- the same
    comparisons and moves as kasa_amd/csrc/stdsort_order.h (checked against std::sort itself in tests/test_host_cpu.py).
    Returns None where libstdc++ would switch to its heap sort (2 log2(n) partitioning levels used up)."""
    a = list(range(n))
    if n <= 1:
        return a
    stack = [(0, n, 2 * (n.bit_length() - 1))]
    while stack:
        first, last, depth = stack.pop()
        while last - first > 16:
            if depth == 0:
                return None
            depth -= 1
            mid = first + (last - first) // 2
            A, B, C = first + 1, mid, last - 1
            if less(a[A], a[B]):
                pick = B if less(a[B], a[C]) else (C if less(a[A], a[C]) else A)
            elif less(a[A], a[C]):
                pick = A
            else:
                pick = C if less(a[B], a[C]) else B
            a[first], a[pick] = a[pick], a[first]
            lo, hi = first + 1, last
            while True:
                while less(a[lo], a[first]):
                    lo += 1
                hi -= 1
                while less(a[first], a[hi]):
                    hi -= 1
                if not lo < hi:
                    break
                a[lo], a[hi] = a[hi], a[lo]
                lo += 1
            stack.append((lo, last, depth))
            last = lo

    def insert_unguarded(i):
        v = a[i]
        j = i - 1
        while less(v, a[j]):
            a[j + 1] = a[j]
            j -= 1
        a[j + 1] = v

    guarded = min(n, 16)
    for i in range(1, guarded):
        if less(a[i], a[0]):
            v = a[i]
            a[1:i + 1] = a[0:i]
            a[0] = v
        else:
            insert_unguarded(i)
    for i in range(guarded, n):
        insert_unguarded(i)
    return a


def rank_read(tax_idx, scores, length: int, freq_khigh, k_high: int, k_low: int, frames: int,
              threshold: float, beasts: int, K: int = K64, protein: bool = False) -> Ranked:
    """Compare.hpp:1495-1594.  `tax_idx` ascending, `scores` > 0 (the cells the reference scans)."""
    best = best_score(length, k_high, k_low, frames, protein)
    thr = float(_F32(threshold))
    hits = []
    for t, s in zip(tax_idx, scores):
        s = _F32(s)
        if not (s > 0):
            continue
        rel = relative_score(s, int(freq_khigh[int(t)]), length, K, protein)
        if rel >= thr:
            hits.append(Hit(int(t), s, rel))
    # std::sort on `rel` descending: not stable beyond 16 elements, but deterministic -- tied hits end up where
    # libstdc++'s introsort leaves them (_stdsort_order); up to 16 elements it is an insertion sort, i.e. stable
    order = _stdsort_order(len(hits), lambda x, y: hits[x].rel > hits[y].rel) if len(hits) > 16 else None
    if order is not None:
        hits = [hits[i] for i in order]
    else:
        hits.sort(key=lambda h: -h.rel)
    n_top = 0
    if hits:
        max_score = max(h.score for h in hits)
        n_top = 1
        for i in range(1, min(len(hits), beasts)):
            if _F32(hits[i].score / max_score) > _F32(0.8):
                n_top += 1
            else:
                break
    return Ranked(hits, n_top, best)


def rank_denominators(freq_khigh, lengths, K: int = K64, protein: bool = False):
    """The host's part of kasa_batch_rank: one row of denominators 1 + log2(freq * span) (Compare.hpp:1506-1511, libm's
    log2 as the reference uses it) per distinct read length.  Returns (den float64[nClasses, nTaxa], class of each read)."""
    lengths = np.asarray(lengths)
    distinct, read_class = np.unique(lengths, return_inverse=True)
    den = np.empty((max(1, distinct.shape[0]), len(freq_khigh)), dtype=np.float64)
    den[:] = np.nan
    for c, length in enumerate(distinct):
        span = (int(length) - (K if protein else 3 * K) + 1) & 0xFFFFFFFF
        for t, f in enumerate(freq_khigh):
            prod = float(int(f)) * float(span)
            lg = math.log2(prod) if prod > 0.0 else (-math.inf if prod == 0.0 else math.nan)
            den[c, t] = 1.0 + lg
    return den, read_class.astype(np.uint32)


def ranked_from_prefix(entries, max_score, length: int, k_high: int, k_low: int, frames: int, beasts: int,
                       protein: bool = False) -> Ranked:
    """A read ranked on the device (kasa_batch_rank): `entries` = the hits a writer can print, in order; `max_score` = the
    largest k-mer score among ALL its hits.  Same top-hit rule as rank_read."""
    best = best_score(length, k_high, k_low, frames, protein)
    hits = [Hit(int(e["tax"]), _F32(e["score"]), float(e["rel"])) for e in entries]
    n_top = 0
    if hits:
        max_score = _F32(max_score)
        n_top = 1
        for i in range(1, min(len(hits), beasts)):
            if _F32(hits[i].score / max_score) > _F32(0.8):
                n_top += 1
            else:
                break
    return Ranked(hits, n_top, best)


def _error(best: np.float32, score: np.float32) -> float:
    return float(_F32(_F32(best - score) / best))


class ReadWriter:
    """Text of the per-read file in one of the reference's four formats (Compare.hpp:1526-1872)."""

    def __init__(self, fmt: str, names, taxids, beasts: int = 3, coherence: bool = False):
        assert fmt in ("json", "jsonl", "tsv", "kraken")
        self.fmt, self.names, self.taxids, self.beasts = fmt, names, taxids, beasts
        self.coherence = coherence      # --coherence: one more column / field (Compare.hpp:1532-1534,1662-1665,1711-1715,1792-1796)
        self.coh = np.float32(0.0)      # ... of the read being written

    def header(self) -> str:
        if self.fmt == "tsv":
            return ("#Read number\tSpecifier from input file\tMatched taxa\tNames\tScores{relative,k-mer}\tError"
                    + ("\tCoherence" if self.coherence else "") + "\n")
        if self.fmt == "json":
            return "[\n"
        return ""

    def footer(self) -> str:
        return "\n]" if self.fmt == "json" else ""

    def _obj(self, h: Hit, best, json_pretty: bool) -> str:
        tid, nm = itoa(self.taxids[h.tax_idx]), self.names[h.tax_idx]
        if json_pretty:
            return ("\t\t\"tax ID\": \"" + tid + "\",\n\t\t\"Name\": \"" + nm + "\",\n\t\t\"k-mer Score\": "
                    + dtoa(float(h.score)) + ",\n\t\t\"Relative Score\": " + dtoa(h.rel)
                    + ",\n\t\t\"Error\": " + dtoa(_error(best, h.score))
                    + ((",\n\t\t\"Coherence\": " + dtoa(float(self.coh))) if self.coherence else "") + "\n\t}")
        return (" \"tax ID\": \"" + tid + "\", \"Name\": \"" + nm + "\", \"k-mer Score\": " + dtoa(float(h.score))
                + ", \"Relative Score\": " + dtoa(h.rel) + ", \"Error\": " + dtoa(_error(best, h.score))
                + ((",\"Coherence\": " + dtoa(float(self.coh))) if self.coherence else "") + "}")

    def _further(self, r: Ranked):
        """Indices printed after the top hits: the -b counter only advances when the k-mer score
        changes (Compare.hpp:1721-1754)."""
        out, j, before = [], r.n_top, _F32(0.0)
        i = r.n_top
        while i < len(r.hits) and j < self.beasts:
            out.append(i)
            if before != r.hits[i].score:
                before = r.hits[i].score
                j += 1
            i += 1
        return out

    def read(self, number: int, name: str, length: int, r: Ranked, coherence=None) -> str:
        f = self.fmt
        self.coh = np.float32(0.0 if coherence is None else coherence)
        if not r.hits:
            if f == "tsv":
                return itoa(number) + "\t" + name + "\t-\t-\t-\t-" + ("\t-" if self.coherence else "") + "\n"
            if f == "json":
                return (("{\n" if number == 0 else ",\n{\n") + "\t\"Read number\": " + itoa(number)
                        + ",\n\t\"Specifier from input file\": \"" + name + "\",\n\t\"Length\": " + itoa(length)
                        + ",\n\t\"Top hits\": [\n\t],\n\t\"Further hits\": [\n\t]\n}")
            if f == "jsonl":
                return ("{ \"Read number\": " + itoa(number) + ", \"Specifier from input file\": \"" + name
                        + "\", \"Length\": " + itoa(length) + ", \"Top hits\": [], \"Further hits\": [] }\n")
            # Kraken: the length is appended as ONE raw byte (Compare.hpp:1568)
            return "U\t" + name + "\t0\t" + chr(length & 0xFF) + "\tA:00\n"
        if f == "tsv":
            s1 = s2 = s3 = s4 = ""
            j, before, i = 0, _F32(0.0), 0
            while i < len(r.hits) and j < self.beasts:
                h = r.hits[i]
                s1 += itoa(self.taxids[h.tax_idx]) + ";"
                s2 += self.names[h.tax_idx] + ";"
                s3 += dtoa(h.rel) + "," + dtoa(float(h.score)) + ";"
                s4 += dtoa(_error(r.best, h.score)) + ";"
                if before != h.score:
                    before = h.score
                    j += 1
                i += 1
            s1, s2, s3, s4 = (x[:-1] if x.endswith(";") else x for x in (s1, s2, s3, s4))
            if not s2:
                return ""
            return (itoa(number) + "\t" + name + "\t" + s1 + "\t" + s2 + "\t" + s3 + "\t" + s4
                    + (("\t" + dtoa(float(self.coh))) if self.coherence else "") + "\n")
        if f == "json":
            out = (("{\n" if number == 0 else ",\n{\n") + "\t\"Read number\": " + itoa(number)
                   + ",\n\t\"Specifier from input file\": \"" + name + "\",\n\t\"Length\": " + itoa(length)
                   + ",\n\t\"Top hits\": [\n")
            for i in range(r.n_top):
                out += ("\t{\n" if i == 0 else ",\n\t{\n") + self._obj(r.hits[i], r.best, True)
            out += "\n\t],\n\t\"Further hits\": [\n"
            for n, i in enumerate(self._further(r)):
                out += ("\t{\n" if n == 0 else ",\n\t{\n") + self._obj(r.hits[i], r.best, True)
            return out + "\n\t]\n}"
        if f == "jsonl":
            out = ("{ \"Read number\": " + itoa(number) + ", \"Specifier from input file\": \"" + name
                   + "\", \"Length\": " + itoa(length) + ", \"Top hits\": [")
            for i in range(r.n_top):
                out += ("{" if i == 0 else ",{") + self._obj(r.hits[i], r.best, False)
            out += "], \"Further hits\": ["
            for n, i in enumerate(self._further(r)):
                out += ("{" if n == 0 else ", {") + self._obj(r.hits[i], r.best, False)
            return out + "] }\n"
        # Kraken-like
        out = ("C\t" + name + "\t" + itoa(self.taxids[r.hits[0].tax_idx]) + "\t" + itoa(length) + "\t")
        for i in list(range(r.n_top)) + self._further(r):
            h = r.hits[i]
            out += itoa(self.taxids[h.tax_idx]) + ":" + dtoa(float(h.score)) + " "
        return out + "\n"


def _g6(x) -> str:
    """operator<< of a double/integer on a default ostream: %g with 6 significant digits."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    return "%g" % float(x)


def _div(a: float, b: float) -> float:
    """IEEE double division as C++ does it (x/0 = inf, 0/0 = nan)."""
    if b == 0.0:
        return math.nan if (a == 0.0 or a != a) else math.copysign(math.inf, a)
    return a / b


def profile_csv(count_all, count_unique, names, taxids, k_high: int, k_low: int, n_kmers_in_input: int,
                n_reads: int, frames: int, count_total=None, freq=None) -> str:
    """Compare.hpp:3466-3665.  Tables are [level, taxon], level 0 = kHigh.  With --coverage pass `count_total`
    (vCount_total) and `freq` (freq[t, l] = k-mers of taxon t at k = kHigh - l, Compare.hpp:166-179): two more
    column groups, "Special Counts" and "Genome Coverage" (Compare.hpp:3574-3581,3627-3637)."""
    nK = k_high - k_low + 1
    n_taxa = len(names)
    sum_u = [int(count_unique[l, 1:].sum()) for l in range(nK)]
    sum_a = [0.0] * nK
    for l in range(nK):
        acc = 0.0
        for t in range(1, n_taxa):
            acc += float(count_all[l, t])
        sum_a[l] = acc
    cov = count_total is not None
    rows = [(names[t].replace(",", " "), [(float(count_all[l, t]), int(count_unique[l, t])) for l in range(nK)],
             int(taxids[t]), t) for t in range(1, n_taxa)]
    rows = [("", [(0.0, 0)] * nK, 0, 0)] + rows  # slot 0 of vOut stays empty (Compare.hpp:3468)

    import functools

    def cmp(a, b):
        for l in range(nK):
            if a[1][l][1] != b[1][l][1]:
                return -1 if a[1][l][1] > b[1][l][1] else 1
        return 0
    rows.sort(key=functools.cmp_to_key(cmp))
    fm = 1 if frames == 1 else (6 if frames == 6 else 3)
    garbage = [0] * nK
    j = 0
    for i in range(k_high - k_low, 0, -1):
        garbage[j] = n_reads * fm * i
        j += 1
    head = "#taxID,Name"
    for title in ("Unique counts k=", "Unique rel. freq. k=", "Non-unique counts k=",
                  "Non-unique rel. freq. k=", "Overall rel. freq. k=", "Overall unique rel. freq. k=") + \
            (("Special Counts k=", "Genome Coverage k=") if cov else ()):
        for l in range(nK):
            head += "," + title + str(k_high - l)
    head += "\n"
    body = ""
    ident = [0.0] * nK
    uident = [0.0] * nK
    for name, vals, tid, tix in rows:
        if not (vals[nK - 1][0] > 0):
            continue
        line = str(tid) + "," + name
        for l in range(nK):
            line += "," + str(vals[l][1])
        for l in range(nK):
            line += "," + ("0" if vals[l][1] == 0 else _g6(vals[l][1] / sum_u[l]))
        for l in range(nK):
            line += "," + _g6(vals[l][0])
        for l in range(nK):
            line += "," + ("0" if vals[l][0] == 0 else _g6(vals[l][0] / sum_a[l]))
        for l in range(nK):
            ident[l] += vals[l][0]
            line += "," + _g6(vals[l][0] / float((n_kmers_in_input - garbage[l]) & 0xFFFFFFFFFFFFFFFF))
        for l in range(nK):
            uident[l] += vals[l][1]
            line += "," + _g6(vals[l][1] / float((n_kmers_in_input - garbage[l]) & 0xFFFFFFFFFFFFFFFF))
        if cov:
            for l in range(nK):
                line += "," + str(int(count_total[l, tix]))
            for l in range(nK):
                line += "," + _g6(_div(float(int(count_total[l, tix])), float(int(freq[tix, l]))))
        body += line + "\n"
    first = "0,not identified" + ",0" * (nK * 4)
    for l in range(nK):
        d = float(n_kmers_in_input) - float(garbage[l])
        first += "," + _g6((d - ident[l]) / d)
    for l in range(nK):
        d = float(n_kmers_in_input) - float(garbage[l])
        first += "," + _g6((d - uident[l]) / d)
    if cov:
        first += ",0" * (2 * nK)
    return head + first + "\n" + body


def is_contaminant(best: np.float32, max_score: np.float32, error_threshold: float = 0.5) -> bool:
    """--filter (Compare.hpp:1597-1599, 2281): a read goes to the contaminants when its best-scoring taxon comes
    within `--errorThreshold` of the perfect score: (best - double(score)) / best < threshold, in double."""
    b = float(np.float32(best))
    return (b - float(np.float32(max_score))) / b < float(np.float32(error_threshold))


def filter_reads(in_paths, flagged, clean_prefix: str, cont_prefix: str, gzip_out: bool = False) -> None:
    """Compare::filter (Compare.hpp:2448-2596): re-read the input file(s) and write every record to `<clean>.fast[aq]`
    or `<contaminants>.fast[aq]` (`_1`/`_2` before the extension for paired input; "_" = do not write that side).
    `flagged`: ascending read numbers of the contaminated reads."""
    import gzip
    def op(path):
        with open(path, "rb") as f:
            magic = f.read(2)
        return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")
    paired = len(in_paths) == 2
    data = []
    for p in in_paths:
        with op(p) as f:
            data.append(f.read())
    fasta = data[0][:1] == b">"
    ext = ".fasta" if fasta else ".fastq"
    def outs(prefix):
        if prefix == "_":
            return None
        names = [prefix + "_1" + ext, prefix + "_2" + ext] if paired else [prefix + ext]
        if gzip_out:                                               # --gzip (Compare.hpp:2455): the same files through zlib, ".gz" appended
            return [gzip.open(n + ".gz", "wb") for n in names]
        return [open(n, "wb") for n in names]
    clean, cont = outs(clean_prefix), outs(cont_prefix)
    try:
        if not flagged and clean is not None:                     # nothing found: the input is copied as it is
            for f, d in zip(clean, data):
                f.write(d)
            return
        lines = []
        for d in data:
            ls = d.split(b"\n")
            if ls and ls[-1] == b"":
                ls.pop()                                           # getline does not return a line after the last '\n'
            lines.append(ls)
        flagged = list(flagged)
        rid = fi = 0
        target = clean
        if fasta:
            for i, l1 in enumerate(lines[0]):
                if l1 == b"":
                    continue
                if l1[:1] == b">":
                    hit = fi < len(flagged) and rid == flagged[fi]
                    target = cont if hit else clean
                    fi += 1 if hit else 0
                    rid += 1
                if target is not None:
                    target[0].write(l1 + b"\n")
                    if paired:
                        target[1].write((lines[1][i] if i < len(lines[1]) else b"") + b"\n")
        else:
            n = len(lines[0])
            i = 0
            while i < n or i == 0:
                rec = [lines[0][i + k] if i + k < n else b"" for k in range(4)]
                rec2 = [lines[1][i + k] if i + k < len(lines[1]) else b"" for k in range(4)] if paired else None
                i += 4
                if rec[0] == b"":
                    if i >= n:
                        break
                    continue
                hit = fi < len(flagged) and rid == flagged[fi]
                target = cont if hit else clean
                fi += 1 if hit else 0
                rid += 1
                if target is not None:
                    target[0].write(b"".join(x + b"\n" for x in rec))
                    if paired:
                        target[1].write(b"".join(x + b"\n" for x in rec2))
                if i >= n:
                    break
    finally:
        for side in (clean, cont):
            for f in side or []:
                f.close()
