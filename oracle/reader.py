"""CPU restatement of how the reference READS its input into batches -- test infrastructure, like the rest of oracle/
(only tests/ import it; the product's hosts have their own code for this: kasa_amd/reads.py, kasa_identify.cpp).

What it follows, in the reference's order (single-end input):

* Utilities::FileReader (source/utils/Utilities.hpp:448-539): the file comes in 2048-byte buffers; getChunk hands out the
  text up to the next line feed or up to the end of the buffer, whichever comes first;
* Read::readFileAndGenerateInfos (source/modes/Read.hpp:371-600): one pass over the file that writes, per record, how
  many lines to skip, how many getChunk calls make up the sequence, and in how many PIECES the sequence is read -- a
  piece ends where the k-mers of what was read so far would take more than 100 MiB of the input vector;
* Read::processInput (Read.hpp:699-760) and its helpers (:612-697): a piece becomes one entry of the batch -- the
  overhang of the piece before it (the last 3K-1 letters) + its own letters + the marker --, all pieces of a record
  under one read id;
* Read::readFastqa_singleEnd (Read.hpp:1054-1233) with strTransfer (:343-356): pieces are taken while more than 100 MiB of
  the -m budget are left; a record whose pieces end up in different batches is an unfinished read ("tail") of the
  first and read 0 of the next;
* Compare::saveResults (source/modes/Compare.hpp:2324-2443): the scores of an unfinished read are kept (taxon, float),
  added to what the next batch finds for it, and printed when its last piece has been scored.

Pinned on tests/golden/batches/long.json, out_long*.jsonl.gz and prof_long*.csv: outputs of the reference binary on a
9.5 Mbp sequence between short reads (tests/golden/make_fixtures.py:case_longseq); the piece lists in long.json are the
binary's own (its temporary file, kept from deletion by tests/golden/batch_probe.c).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

BUFFER = 2048                    # Utilities.hpp:451
PIECE_BYTES = 100 * 1024 * 1024  # Read.hpp:438, 507 (and the floor of the batch budget, Read.hpp:1147)


class FileReader:
    """Utilities::FileReader over the bytes of a file (for .gz: the inflated stream, which is what igzstream::read fills
    the buffer with)."""

    def __init__(self, data: bytes):
        self.data, self.pos, self.eof = data, 0, False

    def get_chunk(self):
        """-> (text, iNumOfChars, ended_with_line_feed or None when nothing could be read)  (Utilities.hpp:514-533)"""
        d, n = self.data, len(self.data)
        if self.pos >= n:               # the refill reads nothing: eofbit stays set (Utilities.hpp:460-474)
            self.eof = True
            return b"", 0, None
        buf_end = (self.pos // BUFFER + 1) * BUFFER
        limit = min(buf_end, n)
        q = d.find(b"\n", self.pos, limit)
        if q >= 0:
            out = (d[self.pos:q], q - self.pos + 1, True)
            self.pos = q + 1
        elif limit == buf_end:          # a full buffer without a line feed
            out = (d[self.pos:limit], limit - self.pos, False)
            self.pos = limit
        else:                           # the last, short buffer: the reader put a line feed behind it (Utilities.hpp:477-481)
            out = (d[self.pos:n], n - self.pos + 1, True)
            self.pos = n
        return out


def kmer_count(length: int, K: int, mode: int) -> int:
    """Read::calculatekMerCount (Read.hpp:36-57).  mode: 0 = DNA in 3 or 6 frames, 1 = --one, 2 = amino acids."""
    if mode == 2:
        return length - K + 1 if length > K + 1 else 0
    if mode == 1:
        t = length // 3
        return t - K + 1 if t > K + 1 else 0
    return length - 3 * K + 1 if length > 3 * K + 1 else 0


def element_bytes(K: int, coherence: bool) -> int:
    """InputType::sizeOf (MetaHeader.h:221-223): tuple<u64, intType, u32, u32> or, with --coherence, the six-field one."""
    return (40 if K > 12 else 32) if coherence else (32 if K > 12 else 24)


def info_lines(data: bytes, fasta: bool, K: int, mode: int, strands: int, coherence: bool = False, piece_bytes: int = PIECE_BYTES):
    """Read::readFileAndGenerateInfos -> the lines of the temporary file as (skip lines, getChunk calls, pieces left)."""
    elem = element_bytes(K, coherence)
    usage = lambda chars: kmer_count(chars, K, mode) * elem * (2 if (strands == 2 and mode != 2) else 1)   # Read.hpp:361-367
    rd = FileReader(data)
    out = []

    def emit(skipped, parts, chunk_no, saved):
        # Read.hpp:398-424 (and :448-466, :515-541): one line per piece, counting the pieces down
        if chunk_no == 1:
            out.append((skipped, parts, 1))
            return
        saved = saved + [parts]
        while chunk_no >= 1:
            out.append((skipped, saved[len(saved) - chunk_no], chunk_no))
            skipped = 0
            chunk_no -= 1

    if fasta:
        skipped = parts = chars = 0
        chunk_no, saved = 0, []
        while not rd.eof:
            text, n, nl = rd.get_chunk()
            if text:
                if text[:1] == b">":
                    emit(skipped, parts, chunk_no, saved)
                    parts, chars, chunk_no, saved = 0, 0, 1, []
                    while not nl:                                   # the rest of the header line
                        _, _, nl = rd.get_chunk()
                        if nl is None:
                            break
                    skipped = 1
                else:
                    parts += 1
                    chars += n
                    if usage(chars) > piece_bytes:
                        chunk_no += 1
                        saved.append(parts)
                        parts = chars = 0
            else:
                parts += 1
        emit(skipped, parts, chunk_no, saved)
        return out
    # FASTQ (Read.hpp:469-598)
    skipped = parts = chars = dna = qual = 0
    chunk_no, saved, kind = 1, [], 0
    while not rd.eof:
        text, n, nl = rd.get_chunk()
        if text:
            if nl:
                n -= 1
            if text[:1] == b"+" and kind == 1:
                kind = 2
            if kind == 0:
                while not nl:
                    _, _, nl = rd.get_chunk()
                    if nl is None:
                        break
                skipped += 1
                kind = 1
            elif kind == 1:
                parts += 1
                chars += n
                dna += n
                if usage(chars) > piece_bytes:
                    chunk_no += 1
                    saved.append(parts)
                    parts = chars = 0
            elif kind == 2:
                emit(skipped, parts, chunk_no, saved)
                parts, chars, chunk_no, saved = 0, 0, 1, []
                while not nl:
                    _, _, nl = rd.get_chunk()
                    if nl is None:
                        break
                skipped = 1
                kind = 3
            else:
                qual += n
                extra = 0
                while not nl:
                    _, extra, nl = rd.get_chunk()
                    if nl is None:
                        break
                    qual += extra
                if extra > 0:
                    qual -= 1
                if qual == dna:
                    dna = qual = 0
                    kind = 0
                if qual > dna:
                    raise RuntimeError("Quality string and DNA string do not have the same length!")
                skipped += 1
        else:
            parts += 1
    out.append((skipped, 0, 0))                                     # Read.hpp:591-598
    return out


@dataclass
class RefBatch:
    """One call of readFastqa_singleEnd."""
    texts: list = field(default_factory=list)       # bytes per entry: overhang + letters (cleaned, padded; without the marker)
    entry_read: list = field(default_factory=list)  # local read id of every entry
    names: list = field(default_factory=list)       # (specifier, Length) of the reads that END in this batch
    n_reads: int = 0                                # iNumOfNewReads: local read ids incl. an unfinished last one
    finished: bool = True                           # strTransfer::finished after the batch
    add_tail: bool = False                          # strTransfer::addTail: the last read goes on in the next batch


def _clean(text: bytes, protein: bool) -> bytes:
    """Read::searchAndReplaceLettersOfRead (Read.hpp:657-675)."""
    if b" " in text or b"\t" in text:
        raise RuntimeError("Spaces or tabs inside read, please check your input.")
    if protein:
        return text.replace(b"*", b"[")
    a = np.frombuffer(text, dtype=np.uint8).copy()
    ok = np.isin(a, np.frombuffer(b"ACGTacgt", dtype=np.uint8))
    a[~ok] = ord("Z")
    return a.tobytes()


def read_batches(data: bytes, fasta: bool, budget: int, K: int, k_low: int, mode: int, strands: int, n_taxa: int,
                 want_per_read: bool = True, coherence: bool = False, piece_bytes: int = PIECE_BYTES, floor_bytes: int = PIECE_BYTES):
    """Read::readFastqa_singleEnd called until the file is used up (Compare.hpp:3091-3135: the budget loses 0.1 % once
    after the first batch) -> list of RefBatch."""
    protein = mode == 2
    unit = 1 if protein else 3
    marker = unit * (K - k_low)                                       # Read.hpp:1068-1078
    elem = element_bytes(K, coherence)
    lines = info_lines(data, fasta, K, mode, strands, coherence, piece_bytes)
    rd = FileReader(data)
    # strTransfer
    t_name, t_overhang, t_length, t_finished, t_line = "", b"", 0, True, 0
    batches, first = [], True
    while True:
        left = budget
        if not first and budget - int(budget * 0.001) > 0:
            left -= int(budget * 0.001)                               # Compare.hpp:3129-3132
        first = False
        b = RefBatch(finished=t_finished)
        line_no, local_id = t_line, 0
        overhang, name, length = t_overhang, t_name, t_length
        add_tail, finished, file_ok = True, t_finished, True
        while True:
            if line_no < len(lines):
                ent = lines[line_no]
                line_no += 1
            else:
                ent, file_ok = (0, 0, 0), False
            if left <= floor_bytes or not file_ok:                   # Read.hpp:1147-1154
                if file_ok:
                    line_no -= 1                                      # the line is read again by the next call
                break
            if ent[2] > 0:                                            # processInput (Read.hpp:699-760)
                text = b""
                for _ in range(ent[0]):
                    text, nl = b"", False
                    while not nl:
                        t, n, nl = rd.get_chunk()
                        if nl is None:
                            break
                        text += t
                if ent[0]:
                    name += text[1:].decode("latin-1") + " "
                parts = []
                for _ in range(ent[1]):
                    t, n, _nl = rd.get_chunk()
                    parts.append(t)
                    length += n
                text = overhang + _clean(b"".join(parts), protein)
                if text:                                              # paddingOfSmallReads (Read.hpp:633-654)
                    if protein:
                        text += b"^" * max(0, K - marker - len(text))
                    elif mode == 1:
                        while (len(text) + marker) // 3 < K:
                            text += b"X"
                    else:
                        text += b"X" * max(0, 3 * K - marker - len(text))
                for _s in range(strands if not protein else 1):       # putReadIntoLocalMemory (Read.hpp:612-630)
                    L = len(text) + marker
                    left -= kmer_count(L, K, mode) * elem + L + 16
                b.texts.append(text)
                b.entry_read.append(local_id)
                if ent[2] == 1:
                    local_id += 1
                if ent[2] > 1:                                        # generateOverhang (Read.hpp:678-697)
                    overhang = text if len(text) < unit * K else text[len(text) + 1 - unit * K:]
                else:
                    overhang = b""
            if ent[2] == 1:                                           # Read.hpp:1160-1174
                finished, add_tail = True, False
                if want_per_read and name != "" and length != 0:
                    left -= 40 + len(name.encode("latin-1")) + 4      # sizeof(pair<string, uint32_t>) = 40
                    b.names.append((name, length & 0xFFFFFFFF))
                    t_name, t_length = "", 0
                name, length = "", 0
            elif ent[2] != 0:                                         # Read.hpp:1175-1186
                finished, add_tail = False, True
                if want_per_read and name != "" and length != 0:
                    t_name = name
                    t_length += length
            if want_per_read and finished:                            # Read.hpp:1189-1194
                left -= (4 if coherence else 0) + n_taxa * 4
        if file_ok:
            t_overhang, t_line = overhang, line_no                    # Read.hpp:1209-1212
        t_finished = finished
        b.n_reads, b.finished, b.add_tail = local_id + int(add_tail), finished, add_tail
        if not b.texts and not file_ok:
            break
        if not b.texts:
            raise RuntimeError("the budget is below the reader's floor: the reference would not get on either")
        batches.append(b)
        if not file_ok:
            break
    return batches


class SavedScores:
    """Compare::saveResults' vSavedScores (Compare.hpp:2324-2443): what an unfinished read has scored so far."""

    def __init__(self):
        self.tax = np.zeros(0, np.uint32)
        self.score = np.zeros(0, np.float32)

    def __bool__(self):
        return self.tax.shape[0] > 0

    def add(self, tax, score):
        """Entries of one more batch; a taxon present on both sides gets the float sum (Compare.hpp:2347-2360, :2391-2405:
        after a sort by taxon there are at most two entries per taxon, so the order of the two does not matter)."""
        tax = np.asarray(tax, np.uint32)
        score = np.asarray(score, np.float32)
        allt = np.union1d(self.tax, tax)
        out = np.zeros(allt.shape[0], np.float32)
        has = np.zeros(allt.shape[0], bool)
        for t, s in ((self.tax, self.score), (tax, score)):
            at = np.searchsorted(allt, t)
            out[at] = np.where(has[at], (out[at] + s).astype(np.float32), s)
            has[at] = True
        self.tax, self.score = allt.astype(np.uint32), out

    def take(self):
        t, s = self.tax, self.score
        self.tax, self.score = np.zeros(0, np.uint32), np.zeros(0, np.float32)
        return t, s
