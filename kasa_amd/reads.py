"""FASTA / FASTQ input as the reference's reader sees it.

Mirrors what reaches the hot path from source/modes/Read.hpp:699-760 (processInput) for inputs whose
reads fit one batch chunk (every read shorter than ~100 MiB of k-mers, i.e. all short-read data):

* the specifier is the header line without its first character **plus a trailing space**
  (Read.hpp:711-714);
* ``Length`` counts one extra per sequence line, because the line feed is part of what the chunk
  reader reports (Read.hpp:723-731): a one-line 150 bp read has length 151;
* the sequence handed on is the concatenation of its lines, untouched -- cleaning, padding and the
  ``X`` marker are applied on the device (kasa_amd/csrc/encode.hip) exactly as Read.hpp:612-675 does.
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass

import numpy as np


@dataclass
class ReadBatch:
    """Reads of one batch: concatenated raw bases + offsets, names and reference-style lengths."""
    bases: np.ndarray     # u8[total]
    offsets: np.ndarray   # i64[n+1]
    names: list           # str, with the trailing space
    lengths: np.ndarray   # u32[n]  ("Length" of the reference's output)
    protein: bool = False  # amino-acid input as kASA::detectAlphabet decides (kASA.hpp:155-183)
    seg_read: np.ndarray = None  # paired-end: u32[nSequences], read of every sequence (offsets delimit sequences then)

    @property
    def n(self) -> int:
        return int(self.lengths.shape[0]) if self.seg_read is not None else int(self.offsets.shape[0] - 1)

    def slice(self, a: int, b: int) -> "ReadBatch":
        names = self.names[a:b] if self.names is not None else None
        if self.seg_read is not None:
            sa, sb = np.searchsorted(self.seg_read, [a, b], side="left")
            o = self.offsets[sa:sb + 1]
            return ReadBatch(self.bases[int(o[0]):int(o[-1])], o - o[0], names, self.lengths[a:b], self.protein,
                             (self.seg_read[sa:sb] - np.uint32(a)).astype(np.uint32))
        o = self.offsets[a:b + 1]
        return ReadBatch(self.bases[int(o[0]):int(o[-1])], o - o[0], names, self.lengths[a:b], self.protein)


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
    names, seqs, lens = [], [], []
    i, n = 0, len(lines)
    if first == b">":
        while i < n:
            if lines[i] == b"":
                i += 1
                continue
            name = lines[i][1:].decode("latin-1") + " "
            i += 1
            parts = []
            while i < n and not lines[i].startswith(b">"):
                if lines[i] != b"":
                    parts.append(lines[i])
                i += 1
            seq = b"".join(parts)
            names.append(name)
            seqs.append(seq)
            lens.append(len(seq) + len(parts))
    else:
        while i < n:
            if lines[i] == b"":
                i += 1
                continue
            name = lines[i][1:].decode("latin-1") + " "
            i += 1
            parts = []
            while i < n and not lines[i].startswith(b"+"):
                parts.append(lines[i])
                i += 1
            seq = b"".join(parts)
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
    return ReadBatch(bases, off, names, np.asarray(lens, dtype=np.uint32), detect_protein(data))


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
