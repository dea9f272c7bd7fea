"""A sequence the reference reads in PIECES and carries across two batches (strTransfer, Read.hpp:343-356; the pieces'
scores merged in Compare.hpp:2344-2426).

tests/golden/batches/long.* (make_fixtures.py:case_longseq) hold what the reference binary made of 2100 short reads, one
9.5 Mbp sequence and 40 more short reads with -m 1, in three and in six frames: the list of pieces it wrote to its
temporary file, its batch sizes, its per-read file and its profile.

* CPU: oracle/reader.py (the restatement of the reader) reproduces the binary's pieces and batches; the oracle run over
  those batches, with the unfinished read's scores carried over, reproduces both files byte for byte;
* GPU: the Python host and the C++ driver write the same bytes.
"""
import gzip
import json
import lzma
import os
import shutil
import subprocess

import numpy as np
import pytest

from kasa_amd import capi, formats, reads
from oracle import reader
from tests import helpers

SRC = os.path.join(helpers.GOLDEN, "batches")
CONFIGS = {"long": 3, "long_six": 6}


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("longseq"))
    for f in ("content.txt.gz", "idx_f.txt.gz"):
        with gzip.open(os.path.join(SRC, f), "rb") as g, open(os.path.join(d, f[:-3]), "wb") as o:
            shutil.copyfileobj(g, o)
    for f in ("idx", "idx_info.txt", "idx_trie", "idx_trie.txt"):
        shutil.copy(os.path.join(SRC, f), os.path.join(d, f))
    with lzma.open(os.path.join(SRC, "long.fasta.xz"), "rb") as g, open(os.path.join(d, "long.fasta"), "wb") as o:
        shutil.copyfileobj(g, o)
    ix = formats.load_index(os.path.join(d, "idx"), os.path.join(d, "content.txt"))
    with open(os.path.join(d, "long.fasta"), "rb") as f:
        data = f.read()
    return d, ix, data, json.load(open(os.path.join(SRC, "long.json")))


def _golden(name):
    with gzip.open(os.path.join(SRC, "out_%s.jsonl.gz" % name), "rb") as f:
        text = f.read().decode("latin-1")
    with open(os.path.join(SRC, "prof_%s.csv" % name), "rb") as f:
        return text, f.read().decode("latin-1")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_oracle_reader_reproduces_the_pieces_and_batches(case, name):
    d, ix, data, gold = case
    frames = CONFIGS[name]
    lines = reader.info_lines(data, True, 12, 0, 2 if frames == 6 else 1)
    long_ones = ["%d,%d,%d" % l for l in lines if l[2] != 1 or l[0] == 0]
    assert long_ones == gold[name]["pieces_of_the_long_sequence"]
    assert len(lines) == 2140 + len(long_ones)
    budget = capi.RefBatcher(ix, 12, 7, frames, memory_gib=1, threads=1).budget
    batches = reader.read_batches(data, True, budget, 12, 7, 0, 2 if frames == 6 else 1, len(ix.content.taxids))
    assert [b.n_reads for b in batches] == gold[name]["batches"]
    assert [b.add_tail for b in batches] == [True, False]
    assert batches[0].entry_read[-2:] == [2099, 2100]                       # the first piece alone ends batch 1 ...
    assert batches[1].entry_read[:len(long_ones)] == [0] * (len(long_ones) - 1) + [1]   # ... the others are read 0 of batch 2


@pytest.mark.parametrize("name,closed_form", [("long", True), ("long", False), ("long_six", True)])
def test_oracle_over_the_pieces_equals_the_reference(case, name, closed_form):
    d, ix, data, gold = case
    frames = CONFIGS[name]
    budget = capi.RefBatcher(ix, 12, 7, frames, memory_gib=1, threads=1).budget
    batches = reader.read_batches(data, True, budget, 12, 7, 0, 2 if frames == 6 else 1, len(ix.content.taxids))
    rows, names, lengths = [], [], []
    saved = reader.SavedScores()
    ca = cu = None
    nq = 0
    for b in batches:
        bases = np.frombuffer(b"".join(b.texts), dtype=np.uint8)
        off = np.zeros(len(b.texts) + 1, np.int64)
        np.cumsum([len(t) for t in b.texts], out=off[1:])
        part = reads.ReadBatch(bases, off, None, np.zeros(b.n_reads, np.uint32), False, np.asarray(b.entry_read, np.uint32))
        res, n = helpers.oracle_identify(ix, part, 12, 7, frames, closed_form=closed_form)
        row = lambda r: (lambda t: (t.astype(np.uint32), res.M[r, t].astype(np.float32)))(np.flatnonzero(res.M[r, 1:] > 0) + 1)
        first = 0
        if saved and b.finished:                                   # Compare.hpp:2344-2386
            saved.add(*row(0))
            rows.append(saved.take())
            first = 1
        last = b.n_reads - 1 if b.add_tail else b.n_reads
        if b.add_tail:                                             # Compare.hpp:2388-2409
            t, s = row(b.n_reads - 1)
            if t.shape[0]:
                saved.add(t, s)
        rows += [row(r) for r in range(first, last)]
        names += [nm for nm, _ in b.names]
        lengths += [ln for _, ln in b.names]
        ca = res.count_all if ca is None else ca + res.count_all
        cu = res.count_unique if cu is None else cu + res.count_unique
        nq += n
        del res
    allr = reads.ReadBatch(None, np.zeros(len(names) + 1, np.int64), names, np.asarray(lengths, np.uint32))
    text, prof = helpers.render(ix, allr, rows, ca, cu, nq, "jsonl", 12, 7, frames, 0.0, 100)
    want_text, want_prof = _golden(name)
    assert prof == want_prof
    assert text == want_text
