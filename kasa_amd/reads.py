"""FASTA / FASTQ input as the reference's reader sees it.

Mirrors what reaches the hot path from source/modes/Read.hpp:699-760 (processInput) for inputs whose
reads fit one batch chunk (every read shorter than ~100 MiB of k-mers, i.e. all short-read data):

* the specifier is the header line without its first character **plus a trailing space**
  (Read.hpp:711-714);
* ``Length`` counts one extra per sequence line, because the line feed is part of what the chunk
  reader reports (Read.hpp:723-731): a one-line 150 bp read has length 151;
* the sequence handed on is the concatenation of its lines, untouched -- cleaning, padding and the
  ``X`` marker are applied on the device (kasa_amd/csrc/encode.hip) exactly as Read.hpp:612-675 does.

A sequence so long that its k-mers would take more than 100 MiB of the reference's input vector (4.4 Mbp in three
frames) is read by the reference in PIECES (Read.hpp:371-600): every piece but the first starts with the last 3K-1
letters of the one before, every piece ends with the marker, and where a piece ends depends on how the 2048-byte
buffers of Utilities::FileReader (Utilities.hpp:448-539) cut the file's lines.  `parse_reads` keeps that cut of a long
record (`ReadBatch.layout`), `ReadBatch.with_pieces` turns it into the pieces the reference scores.
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass

import numpy as np


LONG_SEQUENCE = 1_000_000    # shorter records are one piece whatever the options (the smallest piece: 100 MiB / 48 B / 2 strands)
_BUFFER = 2048               # Utilities.hpp:451
PIECE_BYTES = 100 * 1024 * 1024   # Read.hpp:438,507


@dataclass
class ReadBatch:
    """Reads of one batch: concatenated raw bases + offsets, names and reference-style lengths."""
    bases: np.ndarray     # u8[total]
    offsets: np.ndarray   # i64[n+1]
    names: list           # str, with the trailing space
    lengths: np.ndarray   # u32[n]  ("Length" of the reference's output)
    protein: bool = False  # amino-acid input as kASA::detectAlphabet decides (kASA.hpp:155-183)
    seg_read: np.ndarray = None  # paired-end: u32[nSequences], read of every sequence (offsets delimit sequences then)
    layout: dict = None   # read -> u32[nParts, 2] for records of LONG_SEQUENCE letters and more: (letters, 1 = ended by a line feed) of every getChunk call
    fasta: bool = True
    piece_chars: np.ndarray = None  # with_pieces: what every piece adds to "Length" (its letters + line feeds)

    @property
    def n(self) -> int:
        return int(self.lengths.shape[0]) if self.seg_read is not None else int(self.offsets.shape[0] - 1)

    def with_pieces(self, K: int, frames: int, coherence: bool = False, piece_bytes: int = PIECE_BYTES) -> "ReadBatch":
        """The batch as the reference reads it: a record it cuts into pieces (module text) becomes several sequences of one
        read -- every piece but the first starts with the overhang of the one before (Read.hpp:678-697, 738-741) --,
        `seg_read` names the read of every sequence and `piece_chars` what the piece adds to the read's "Length".
        Returns self when no record is cut."""
        if not self.layout:
            return self
        if self.seg_read is not None:
            raise RuntimeError("a paired-end input with sequences long enough for kASA to read them in pieces is not supported")
        mode = 2 if self.protein else (1 if frames == 1 else 0)
        strands = 2 if (frames == 6 and not self.protein) else 1
        over = (K if self.protein else 3 * K) - 1
        cuts = {}
        for r, parts in self.layout.items():
            c, add = piece_cuts(parts, self.fasta, K, mode, strands, coherence, piece_bytes)
            if len(c) > 2:
                cuts[r] = (c, add)
        if not cuts:
            return self
        chunks, off, seg, chars = [], [0], [], []
        for r in range(self.n):
            a, b = int(self.offsets[r]), int(self.offsets[r + 1])
            if r not in cuts:
                pieces, adds = [self.bases[a:b]], [int(self.lengths[r])]
            else:
                c, adds = cuts[r]
                pieces, text_len = [], 0
                for i in range(len(c) - 1):
                    keep = min(over, text_len)                      # generateOverhang: the last 3K-1 letters of the text before, or all of it
                    pieces.append(self.bases[a + c[i] - keep:a + c[i + 1]])
                    text_len = keep + c[i + 1] - c[i]
            for pc, add in zip(pieces, adds):
                chunks.append(pc)
                off.append(off[-1] + pc.shape[0])
                seg.append(r)
                chars.append(add)
        return ReadBatch(np.concatenate(chunks) if chunks else np.zeros(0, np.uint8), np.asarray(off, np.int64), self.names, self.lengths,
                         self.protein, np.asarray(seg, np.uint32), None, self.fasta, np.asarray(chars, np.int64))

    def slice(self, a: int, b: int) -> "ReadBatch":
        names = self.names[a:b] if self.names is not None else None
        if self.seg_read is not None:
            sa, sb = np.searchsorted(self.seg_read, [a, b], side="left")
            o = self.offsets[sa:sb + 1]
            return ReadBatch(self.bases[int(o[0]):int(o[-1])], o - o[0], names, self.lengths[a:b], self.protein,
                             (self.seg_read[sa:sb] - np.uint32(a)).astype(np.uint32))
        o = self.offsets[a:b + 1]
        return ReadBatch(self.bases[int(o[0]):int(o[-1])], o - o[0], names, self.lengths[a:b], self.protein)


def _kmer_count(length: int, K: int, mode: int) -> int:
    """Read.hpp:36-57.  mode: 0 = DNA in 3 or 6 frames, 1 = --one, 2 = amino acids."""
    if mode == 2:
        return length - K + 1 if length > K + 1 else 0
    if mode == 1:
        return length // 3 - K + 1 if length // 3 > K + 1 else 0
    return length - 3 * K + 1 if length > 3 * K + 1 else 0


def _chunk_parts(data: bytes, begin: int, end: int) -> np.ndarray:
    """The getChunk calls (Utilities.hpp:514-533) that cover data[begin, end), a run of whole lines that starts where a
    call starts: every call ends at a line feed or at the next multiple of 2048 bytes in the file."""
    parts = []
    pos, n = begin, len(data)
    while pos < end:
        limit = min((pos // _BUFFER + 1) * _BUFFER, n)
        q = data.find(b"\n", pos, limit)
        if q >= 0:
            parts.append((q - pos, 1))
            pos = q + 1
        elif limit < n or n % _BUFFER == 0:
            parts.append((limit - pos, 0))
            pos = limit
        else:                                  # the file's last line has no line feed: the reader supplies one (Utilities.hpp:477-481)
            parts.append((n - pos, 1))
            pos = n
    return np.asarray(parts, dtype=np.uint32).reshape(-1, 2)


def piece_cuts(parts: np.ndarray, fasta: bool, K: int, mode: int, strands: int, coherence: bool = False,
               piece_bytes: int = PIECE_BYTES):
    """Where Read::readFileAndGenerateInfos ends the pieces of one record (Read.hpp:434-443, 503-512): after the getChunk
    call with which the k-mers of the letters (FASTA: and line feeds) read so far pass `piece_bytes` of the input vector.
    -> (letters before every cut, [0, ..., all]; what every piece adds to "Length")."""
    elem = (40 if K > 12 else 32) if coherence else (32 if K > 12 else 24)      # InputType::sizeOf, MetaHeader.h:221-223
    mult = elem * (2 if (strands == 2 and mode != 2) else 1)                    # Read.hpp:361-367
    # the smallest count of characters whose k-mers pass the limit
    lo, hi = 0, 1 << 40
    while lo + 1 < hi:
        mid = (lo + hi) // 2
        if _kmer_count(mid, K, mode) * mult > piece_bytes:
            hi = mid
        else:
            lo = mid
    need = hi
    letters = parts[:, 0].astype(np.int64)
    feeds = parts[:, 1].astype(np.int64)
    counted = np.where(letters > 0, letters + (feeds if fasta else 0), 0)       # (a call that returns no text is not counted: Read.hpp:394,445)
    cl, cc, ct = np.cumsum(letters), np.cumsum(counted), np.cumsum(letters + feeds)   # Read.hpp:723-731: "Length" counts the line feeds
    cuts, adds, base_c, base_t = [0], [], 0, 0
    while True:
        i = int(np.searchsorted(cc, base_c + need, side="left"))               # the call with which the count reaches `need`
        if i >= len(cc):
            break
        cuts.append(int(cl[i]))
        adds.append(int(ct[i]) - base_t)
        base_c, base_t = int(cc[i]), int(ct[i])
    cuts.append(int(cl[-1]))                                                    # the last piece: what is left, possibly nothing (Read.hpp:411-421)
    adds.append(int(ct[-1]) - base_t)
    return cuts, adds


def _open(path: str):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


_DNA4 = re.compile(r"^[ACGTURYKMSWBDHVN-]+$", re.IGNORECASE)


def detect_protein(data: bytes) -> bool:
    """kASA::detectAlphabet (kASA.hpp:155-183) on what Utilities::getFirstSequenceOfFile hands it
    (Utilities.hpp:137-143): the first four characters of the file's second line.  Anything that is not
    made of IUPAC nucleotide letters counts as amino-acid input."""
    lines = data.split(b"\n", 2)
    second = lines[1] if len(lines) > 1 else b""
    return _DNA4.match(second[:4].decode("latin-1")) is None


def parse_reads(path: str) -> ReadBatch:
    """Whole file -> ReadBatch.  Raises like Compare.hpp:2984-2994 on an unknown first character."""
    with _open(path) as f:
        data = f.read()
    if not data:
        return ReadBatch(np.zeros(0, np.uint8), np.zeros(1, np.int64), [], np.zeros(0, np.uint32))
    first = data[:1]
    if first not in (b">", b"@"):
        raise RuntimeError("Input does not start with @ or >.")
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    names, seqs, lens, span = [], [], [], []
    i, n = 0, len(lines)
    if first == b">":
        while i < n:
            if lines[i] == b"":
                i += 1
                continue
            name = lines[i][1:].decode("latin-1") + " "
            i += 1
            parts, i0 = [], i
            while i < n and not lines[i].startswith(b">"):
                if lines[i] != b"":
                    parts.append(lines[i])
                i += 1
            seq = b"".join(parts)
            names.append(name)
            seqs.append(seq)
            lens.append(len(seq) + len(parts))
            span.append((i0, i))
    else:
        while i < n:
            if lines[i] == b"":
                i += 1
                continue
            name = lines[i][1:].decode("latin-1") + " "
            i += 1
            parts, i0 = [], i
            while i < n and not lines[i].startswith(b"+"):
                parts.append(lines[i])
                i += 1
            seq = b"".join(parts)
            span.append((i0, i))
            i += 1  # '+' line
            q = 0
            while i < n and q < len(seq):  # quality: as many characters as bases
                q += len(lines[i])
                i += 1
            if q > len(seq):
                raise RuntimeError("Quality string and DNA string do not have the same length!")
            names.append(name)
            seqs.append(seq)
            lens.append(len(seq) + len(parts))
    for s in seqs:
        if b" " in s or b"\t" in s:
            raise RuntimeError("Spaces or tabs inside read, please check your input.")  # Read.hpp:659-661
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    if seqs:
        np.cumsum([len(s) for s in seqs], out=off[1:])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    layout = None
    long_ones = [r for r, s in enumerate(seqs) if len(s) >= LONG_SEQUENCE]
    if long_ones:                       # how the reference's reader cuts these records' lines (see the module text)
        starts = np.concatenate([[0], np.cumsum(np.fromiter((len(l) + 1 for l in lines), np.int64, len(lines)))])
        layout = {r: _chunk_parts(data, int(starts[span[r][0]]), min(int(starts[span[r][1]]), len(data))) for r in long_ones}
    return ReadBatch(bases, off, names, np.asarray(lens, dtype=np.uint32), detect_protein(data), layout=layout, fasta=first == b">")


def parse_pairs(path1: str, path2: str) -> ReadBatch:
    """-1 <file> -2 <file>: mate i of both files forms read i (Read.hpp:834-1049).  Both sequences keep their own
    k-mers under one read id; the specifier is both names (each with its trailing space), the length the sum."""
    b1, b2 = parse_reads(path1), parse_reads(path2)
    if b1.n != b2.n:
        raise RuntimeError("paired-end files hold different numbers of reads")
    n = b1.n
    parts, off = [], [0]
    for r in range(n):
        for b in (b1, b2):
            parts.append(b.bases[int(b.offsets[r]):int(b.offsets[r + 1])])
            off.append(off[-1] + parts[-1].shape[0])
    bases = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return ReadBatch(bases, np.asarray(off, dtype=np.int64), [b1.names[r] + b2.names[r] for r in range(n)],
                     (b1.lengths + b2.lengths).astype(np.uint32), b1.protein,
                     np.repeat(np.arange(n, dtype=np.uint32), 2))


def synthetic_reads(genomes, n_reads: int, read_len: int, seed: int, sub_rate: float = 0.01) -> ReadBatch:
    """Seeded synthetic reads: uniform positions on uniform taxa, `sub_rate` substitutions
    (SURVEY.md section 8(d)).  `genomes` is a list of u8 arrays."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, len(genomes), size=n_reads)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    for gi in range(len(genomes)):
        sel = np.flatnonzero(g == gi)
        if sel.size == 0:
            continue
        gen = genomes[gi]
        pos = rng.integers(0, gen.shape[0] - read_len + 1, size=sel.size)
        out[sel] = gen[pos[:, None] + np.arange(read_len)[None, :]]
    mut = rng.random((n_reads, read_len)) < sub_rate
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    out[mut] = alphabet[rng.integers(0, 4, size=int(mut.sum()))]
    off = np.arange(n_reads + 1, dtype=np.int64) * read_len
    names = [f"r{i} " for i in range(n_reads)]
    return ReadBatch(out.reshape(-1), off, names, np.full(n_reads, read_len + 1, dtype=np.uint32))
