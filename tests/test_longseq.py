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
CONFIGS = {"long": ("long.fasta", 3), "long_six": ("long.fasta", 6), "long2_six": ("long2.fasta", 6), "long3": ("long3.fastq", 3)}   # name -> (input, frames)


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("longseq"))
    for f in ("content.txt.gz", "idx_f.txt.gz"):
        with gzip.open(os.path.join(SRC, f), "rb") as g, open(os.path.join(d, f[:-3]), "wb") as o:
            shutil.copyfileobj(g, o)
    for f in ("idx", "idx_info.txt", "idx_trie", "idx_trie.txt"):
        shutil.copy(os.path.join(SRC, f), os.path.join(d, f))
    data = {}
    for stem in ("long.fasta", "long2.fasta", "long3.fastq"):
        with lzma.open(os.path.join(SRC, stem + ".xz"), "rb") as g:
            data[stem] = g.read()
        with open(os.path.join(d, stem), "wb") as o:
            o.write(data[stem])
    ix = formats.load_index(os.path.join(d, "idx"), os.path.join(d, "content.txt"))
    return d, ix, data, json.load(open(os.path.join(SRC, "long.json")))


def _golden(name):
    with gzip.open(os.path.join(SRC, "out_%s.jsonl.gz" % name), "rb") as f:
        text = f.read().decode("latin-1")
    with open(os.path.join(SRC, "prof_%s.csv" % name), "rb") as f:
        return text, f.read().decode("latin-1")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_oracle_reader_reproduces_the_pieces_and_batches(case, name):
    d, ix, data, gold = case
    stem, frames = CONFIGS[name]
    data, fasta = data[stem], stem.endswith(".fasta")
    lines = reader.info_lines(data, fasta, 12, 0, 2 if frames == 6 else 1)
    long_ones = ["%d,%d,%d" % l for l in lines if l[2] > 1 or (l[0] == 0 and l[2] == 1)]
    assert long_ones == gold[name]["pieces_of_the_long_sequence"]
    if fasta:
        assert len(lines) == data.count(b">") - data.count(b">sequence") - data.count(b">contig") + len(long_ones)
    else:
        assert len(lines) == 2140 + len(long_ones) + 1 and lines[-1] == (2, 0, 0)      # the file's end: quality line and nothing (Read.hpp:591-598)
    budget = capi.RefBatcher(ix, 12, 7, frames, memory_gib=1, threads=1).budget
    batches = reader.read_batches(data, fasta, budget, 12, 7, 0, 2 if frames == 6 else 1, len(ix.content.taxids))
    assert [b.n_reads for b in batches] == gold[name]["batches"]
    if stem != "long2.fasta":
        assert [b.add_tail for b in batches] == [True, False]
        assert batches[0].entry_read[-2:] == [2099, 2100]                       # the first piece alone ends batch 1 ...
        assert batches[1].entry_read[:len(long_ones)] == [0] * (len(long_ones) - 1) + [1]   # ... the others are read 0 of batch 2
    else:
        # batch 2 begins AND ends inside a sequence, batch 3 lies inside one altogether
        assert [(b.entry_read[0] == 0 and i > 0, b.add_tail) for i, b in enumerate(batches)] == [(False, True), (True, True), (True, True), (True, False)]
        assert len(batches[2].texts) == 9 and batches[2].n_reads == 1


def _oracle_over_reference_batches(case, name, closed_form):
    """-> (rows, names, lengths, countAll, countUnique, nQueries) of the oracle run over the restated reader's batches, the
    unfinished reads' scores carried over as Compare::saveResults does."""
    d, ix, data, gold = case
    stem, frames = CONFIGS[name]
    data, fasta = data[stem], stem.endswith(".fasta")
    budget = capi.RefBatcher(ix, 12, 7, frames, memory_gib=1, threads=1).budget
    batches = reader.read_batches(data, fasta, budget, 12, 7, 0, 2 if frames == 6 else 1, len(ix.content.taxids))
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
    return rows, names, lengths, ca, cu, nq


@pytest.mark.parametrize("name,closed_form", [("long", True), ("long", False), ("long_six", True), ("long2_six", True), ("long3", True)])
def test_oracle_over_the_pieces_equals_the_reference(case, name, closed_form):
    d, ix, data, gold = case
    frames = CONFIGS[name][1]
    rows, names, lengths, ca, cu, nq = _oracle_over_reference_batches(case, name, closed_form)
    allr = reads.ReadBatch(None, np.zeros(len(names) + 1, np.int64), names, np.asarray(lengths, np.uint32))
    text, prof = helpers.render(ix, allr, rows, ca, cu, nq, "jsonl", 12, 7, frames, 0.0, 100)
    want_text, want_prof = _golden(name)
    assert prof == want_prof
    assert text == want_text


def test_oracle_over_the_pieces_tsv_and_filter(case, tmp_path):
    """The same input through the TSV writer (-b 3) with --filter: the read finished from what batch 1 left of it takes the
    reference's one-read writer and filter test (Compare.hpp:1894-2265); its files are the fixture (the filter's two files by
    their SHA-256: 10 MB each)."""
    import hashlib
    from kasa_amd import report
    d, ix, data, gold = case
    rows, names, lengths, ca, cu, nq = _oracle_over_reference_batches(case, "long", True)
    allr = reads.ReadBatch(None, np.zeros(len(names) + 1, np.int64), names, np.asarray(lengths, np.uint32))
    text, prof = helpers.render(ix, allr, rows, ca, cu, nq, "tsv", 12, 7, 3, 0.0, 3)
    with gzip.open(os.path.join(SRC, "out_long_flt.tsv.gz"), "rb") as f:
        assert text == f.read().decode("latin-1")
    flagged = []
    for r in range(len(rows)):
        rk = report.rank_read(rows[r][0], rows[r][1], int(lengths[r]), ix.freq_at(12), 12, 7, 3, 0.0, 3)
        if rk.hits and report.is_contaminant(rk.best, max(h.score for h in rk.hits), 0.5):
            flagged.append(r)
    report.filter_reads([os.path.join(d, "long.fasta")], flagged, str(tmp_path / "c"), str(tmp_path / "x"))
    for mine, ref in (("c.fasta", "lflt_clean.fasta"), ("x.fasta", "lflt_cont.fasta")):
        with open(str(tmp_path / mine), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == gold["long_flt"]["sha256"][ref], ref


@pytest.mark.parametrize("name", list(CONFIGS))
def test_host_pieces_and_batches_equal_the_oracle_reader(case, name):
    """kasa_amd/reads.py + capi.RefBatcher.piece_batches (the product's own code for this) against oracle/reader.py."""
    d, ix, data, gold = case
    stem, frames = CONFIGS[name]
    data, fasta = data[stem], stem.endswith(".fasta")
    batch = reads.parse_reads(os.path.join(d, stem))
    pieced = batch.with_pieces(12, frames)
    rb = capi.RefBatcher(ix, 12, 7, frames, memory_gib=1, threads=1)
    bounds = rb.piece_batches(pieced, True)
    want = reader.read_batches(data, fasta, rb.budget, 12, 7, 0, 2 if frames == 6 else 1, len(ix.content.taxids))
    assert len(bounds) - 1 == len(want)
    for (pa, pb), b in zip(zip(bounds[:-1], bounds[1:]), want):
        assert pb - pa == len(b.texts)
        for q in range(pa, pb):
            assert pieced.bases[int(pieced.offsets[q]):int(pieced.offsets[q + 1])].tobytes() == b.texts[q - pa]
        assert list(pieced.seg_read[pa:pb] - pieced.seg_read[pa]) == b.entry_read
    for r in batch.layout:
        assert int(pieced.piece_chars[pieced.seg_read == r].sum()) == int(batch.lengths[r])
    assert batch.with_pieces(12, frames, piece_bytes=1 << 40) is batch        # nothing to cut: the batch itself


def test_pieces_of_small_inputs_follow_the_oracle_reader():
    """Line lengths, buffer ends and record ends in every position against each other, with a piece limit of a few hundred
    k-mers: reads.py's cuts against the oracle reader's list of pieces (FASTA and FASTQ, all three k-mer geometries)."""
    rng = np.random.default_rng(5)
    for trial in range(60):
        fasta = trial % 2 == 0
        width = int(rng.integers(20, 3000))
        recs = []
        for r in range(int(rng.integers(1, 5))):
            n = int(rng.integers(1, 9000))
            seq = "".join(rng.choice(list("ACGT"), n))
            body = "\n".join(seq[i:i + width] for i in range(0, n, width))
            tag = "x" * int(rng.integers(0, 40) if trial % 4 else rng.integers(1500, 5000))   # (a header longer than the reader's buffer)
            recs.append((">r%d %s\n%s\n" % (r, tag, body)) if fasta else
                        ("@r%d %s\n%s\n+\n%s\n" % (r, tag, body, "\n".join("I" * len(l) for l in body.split("\n")))))
        data = "".join(recs).encode()
        if trial % 5 == 4:
            data = data[:-1]                                                    # no line feed at the end
        if trial % 7 == 3 and fasta:
            data = data.replace(b"\n", b"\r\n")                                # the '\r' stays a letter of its line
        mode, strands = [(0, 1), (0, 2), (1, 1)][trial % 3]
        limit = int(rng.integers(2000, 60000))
        lines = reader.info_lines(data, fasta, 12, mode, strands, piece_bytes=limit)
        want = {}
        rec = -1
        for skip, parts, left in lines:
            if skip:
                rec += 1
            if left:
                want.setdefault(rec, []).append(parts)
        # the product's parts and cuts of every record
        text = data.split(b"\n")
        starts = np.concatenate([[0], np.cumsum([len(l) + 1 for l in text])])
        i, rec = 0, -1
        while i < len(text):
            if text[i][:1] not in (b">", b"@"):
                i += 1
                continue
            rec += 1
            i0 = i + 1
            i = i0
            while i < len(text) and text[i][:1] != (b">" if fasta else b"+") and not (i == len(text) - 1 and text[i] == b""):
                i += 1
            parts = reads._chunk_parts(data, int(starts[i0]), min(int(starts[i]), len(data)))
            cuts, adds = reads.piece_cuts(parts, fasta, 12, mode, strands, piece_bytes=limit)
            cl = np.concatenate([[0], np.cumsum(parts[:, 0])])
            calls = [int(np.searchsorted(cl, c, side="left")) if k else 0 for k, c in enumerate(cuts)]
            calls[-1] = len(parts)
            got = list(np.diff(calls))
            ref = want[rec]
            if rec == len(recs) - 1 and fasta:
                ref = ref[:-1] + [ref[-1] - 1]          # the call at the end of the file that returns nothing (Read.hpp:445)
            assert got == ref, (trial, rec, got, ref)
            if not fasta:                               # skip the '+' and quality lines
                i += 1
                while i < len(text) and text[i][:1] != b"@":
                    i += 1


# ------------------------------------------------------------------------------------------------ GPU
# (one wavefront replays a read's events in their order: a sequence of 20 M k-mers takes the device half a minute -- DESIGN.md
# section 8 --, so the GPU cases are a selection: every input through both hosts once)
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["long2_six", "long3"])
def test_python_host_byte_identical_over_the_pieces(case, name):
    from kasa_amd import identify
    d, ix, data, gold = case
    stem, frames = CONFIGS[name]
    data, fasta = data[stem], stem.endswith(".fasta")
    batch = reads.parse_reads(os.path.join(d, stem))
    run = identify.Identify(ix, 0, 12, 7, frames, 0.0, 100, "jsonl")
    text, prof, _ = run.run(batch, True, memory_gib=1, threads=1)
    assert run.batch_sizes == gold[name]["batches"]
    want_text, want_prof = _golden(name)
    assert prof == want_prof
    assert text == want_text                                 # scores of the long sequence included, nothing stripped
    # without per-read output the pieces still decide the k-mers (every piece ends with the marker): same profile
    _, prof_only, _ = run.run(batch, False)
    assert prof_only == want_prof
    run.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra", [("long2_six", ["gz", "KASA_READ_BLOCK=5000000", "KASA_PARSE_CHUNK=200000"]), ("long3", [])])
def test_cpp_host_byte_identical_over_the_pieces(case, name, extra, tmp_path):
    """kasa_identify streams the file in blocks and parses them with several threads: where the blocks and the threads' runs
    end must not move the pieces (the reader's 2048-byte buffers are counted from the start of the file)."""
    from kasa_amd import build as hipbuild
    d, ix, data, gold = case
    stem, frames = CONFIGS[name]
    data, fasta = data[stem], stem.endswith(".fasta")
    exe = hipbuild.build_host()
    src = os.path.join(d, stem)
    env = dict(os.environ)
    if "gz" in extra:
        with open(src, "rb") as f, gzip.open(str(tmp_path / ("in." + stem.rsplit(".", 1)[1] + ".gz")), "wb", compresslevel=1) as g:
            shutil.copyfileobj(f, g)
        src = str(tmp_path / ("in." + stem.rsplit(".", 1)[1] + ".gz"))
        env.update(e.split("=") for e in extra if "=" in e)
    out, prof = str(tmp_path / "out.jsonl"), str(tmp_path / "prof.csv")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", src, "-q", out, "-p", prof,
           "--jsonl", "-b", "100", "-m", "1", "-n", "1", "-v"] + (["--six"] if frames == 6 else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    sizes = [int(l.split()[3]) for l in r.stdout.splitlines() if l.startswith("OUT: Batch of")]
    assert sizes == gold[name]["batches"], r.stdout
    assert "the last goes on in the next" in r.stdout and "the first goes on from the batch before" in r.stdout
    want_text, want_prof = _golden(name)
    with open(prof, "rb") as f:
        assert f.read().decode("latin-1") == want_prof
    with open(out, "rb") as f:
        assert f.read().decode("latin-1") == want_text
    # a profile-only run reads the same pieces
    prof2 = str(tmp_path / "prof2.csv")
    cmd2 = list(cmd)
    del cmd2[cmd2.index("-q"):cmd2.index("-q") + 2]
    cmd2[cmd2.index(prof)] = prof2
    r = subprocess.run(cmd2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(prof2, "rb") as f:
        assert f.read().decode("latin-1") == want_prof


@pytest.mark.parametrize("block,run", [(None, None), ("1500", "400"), ("4099", "777")])
def test_cpp_driver_cuts_pieces_like_the_python_host(block, run, tmp_path, monkeypatch):
    """kasa_identify's own cutting code (parseRecords' reader calls, readerPieces) without a device: its `pieces-dump` tap over
    small files with every record "long" and a piece limit of a few hundred k-mers, streamed in small blocks and parsed in
    small runs, against kasa_amd/reads.py (which the test above holds against the restated reader)."""
    from kasa_amd import build as hipbuild
    exe = hipbuild.HOST_BIN
    if not os.path.exists(exe):
        pytest.skip("host driver not built (python -m kasa_amd.build)")
    monkeypatch.setattr(reads, "LONG_SEQUENCE", 1)
    rng = np.random.default_rng(9)
    for trial in range(24):
        fasta = trial % 2 == 0
        width = int(rng.integers(20, 3000))
        recs = []
        for r in range(int(rng.integers(1, 6))):
            n = int(rng.integers(40, 12000))
            seq = "".join(rng.choice(list("ACGT"), n))
            body = "\n".join(seq[i:i + width] for i in range(0, n, width))
            tag = "x" * int(rng.integers(0, 40) if trial % 4 else rng.integers(1500, 5000))
            recs.append((">r%d %s\n%s\n" % (r, tag, body)) if fasta else
                        ("@r%d %s\n%s\n+\n%s\n" % (r, tag, body, "\n".join("I" * len(l) for l in body.split("\n")))))
        data = "".join(recs).encode()
        if trial % 5 == 4:
            data = data[:-1]
        if trial % 7 == 3 and fasta:
            data = data.replace(b"\n", b"\r\n")
        path = str(tmp_path / ("t%d.%s" % (trial, "fasta" if fasta else "fastq")))
        with open(path, "wb") as f:
            f.write(data)
        frames = [3, 6, 1][trial % 3]
        limit = int(rng.integers(2000, 60000))
        env = dict(os.environ, KASA_LONG_SEQUENCE="1", KASA_PIECE_BYTES=str(limit))
        if block:
            env.update(KASA_READ_BLOCK=block, KASA_PARSE_CHUNK=run)
        r = subprocess.run([exe, "pieces-dump", path, "3", str(frames), "12"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60, env=env)
        assert r.returncode == 0, r.stderr[-400:]
        got = {int(l.split("\t")[0]): l.split("\t")[1:] for l in r.stdout.splitlines() if l[:1].isdigit()}
        batch = reads.parse_reads(path)
        assert sorted(got) == sorted(batch.layout)
        for rd, parts in batch.layout.items():
            cuts, adds = reads.piece_cuts(parts, fasta, 12, 1 if frames == 1 else 0, 2 if frames == 6 else 1, piece_bytes=limit)
            assert got[rd] == [",".join(map(str, cuts)), ",".join(map(str, adds))], (trial, rd)


def _oracle_over_pieces(ix, pieced, bounds, frames, fmt="jsonl"):
    """The oracle over explicit batches of pieces with Compare::saveResults' rules and the reader's "Length" (test-side restatement,
    the one test_oracle_over_the_pieces_equals_the_reference pins on the reference's files)."""
    seg = pieced.seg_read.astype(np.int64)
    n_pieces = len(seg)
    rows, names, lengths = [], [], []
    saved = reader.SavedScores()
    ca = cu = None
    nq = carried = 0
    for pa, pb in zip(bounds[:-1], bounds[1:]):
        r0 = int(seg[pa])
        local = (seg[pa:pb] - r0).astype(np.uint32)
        n_local = int(local[-1]) + 1
        tail = pb < n_pieces and seg[pb] == seg[pb - 1]
        o = pieced.offsets[pa:pb + 1]
        part = reads.ReadBatch(pieced.bases[int(o[0]):int(o[-1])], o - o[0], None, np.zeros(n_local, np.uint32), pieced.protein, local)
        res, n = helpers.oracle_identify(ix, part, 12, 7, frames, closed_form=True)
        row = lambda r: (lambda t: (t.astype(np.uint32), res.M[r, t].astype(np.float32)))(np.flatnonzero(res.M[r, 1:] > 0) + 1)
        length = carried                                          # Read.hpp:1117,1166-1186
        for q in range(pa, pb):
            length += int(pieced.piece_chars[q])
            if q + 1 == n_pieces or seg[q + 1] != seg[q]:
                names.append(pieced.names[int(seg[q])])
                lengths.append(length & 0xFFFFFFFF)
                length = carried = 0
            else:
                carried += length
        first = 0
        if saved and not tail:                                    # Compare.hpp:2344-2386
            saved.add(*row(0))
            rows.append(saved.take())
            first = 1
        if tail:                                                  # Compare.hpp:2388-2409
            t, s = row(n_local - 1)
            if t.shape[0]:
                saved.add(t, s)
        rows += [row(r) for r in range(first, n_local - (1 if tail else 0))]
        ca = res.count_all if ca is None else ca + res.count_all
        cu = res.count_unique if cu is None else cu + res.count_unique
        nq += n
    allr = reads.ReadBatch(None, np.zeros(len(names) + 1, np.int64), names, np.asarray(lengths, np.uint32))
    return helpers.render(ix, allr, rows, ca, cu, nq, fmt, 12, 7, frames, 0.0, 100)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("KASA_LONGSEQ_SEED0", "0")), int(os.environ.get("KASA_LONGSEQ_SEED0", "0")) + int(os.environ.get("KASA_LONGSEQ_SEEDS", "24"))))      # (more seeds beyond the suite: DESIGN.md section 7)
def test_batches_that_end_inside_reads_three_ways(seed, tmp_path, monkeypatch):
    """Random inputs over the small golden index with every record "long", a piece limit of a few hundred k-mers and device
    batches of a few thousand: batches end inside reads in every constellation (several unfinished reads in a row, a batch
    that is one piece, a read that finishes where the next begins to be unfinished, pieces without a match).  The C++ driver,
    the Python host and the oracle over the same pieces must write the same bytes."""
    from kasa_amd import build as hipbuild, identify
    d, ix = helpers.load_case("pairs")
    genomes = [l.strip() for l in open(os.path.join(d, "db.fasta")) if not l.startswith(">")]
    monkeypatch.setattr(reads, "LONG_SEQUENCE", 1)
    rng = np.random.default_rng(100 + seed)
    frames = [3, 6, 1][seed % 3]
    fasta = seed % 2 == 0
    width = int(rng.integers(50, 400))
    db = "".join(genomes)
    recs = []
    for r in range(int(rng.integers(6, 14))):
        n = int(rng.integers(60, 9000)) if rng.random() < 0.7 else int(rng.integers(30, 200))
        if rng.random() < 0.75:
            a = int(rng.integers(0, max(1, len(db) - n)))
            seq = list(db[a:a + n])
            for i in np.flatnonzero(rng.random(len(seq)) < 0.02):
                seq[i] = "ACGT"[int(rng.integers(0, 4))]
            seq = "".join(seq)
        else:
            seq = "".join(rng.choice(list("ACGT"), n))                # matches next to nothing
        body = "\n".join(seq[i:i + width] for i in range(0, len(seq), width))
        recs.append((">r%d\n%s\n" % (r, body)) if fasta else ("@r%d\n%s\n+\n%s\n" % (r, body, "\n".join("I" * len(l) for l in body.split("\n")))))
    path = str(tmp_path / ("in.fasta" if fasta else "in.fastq"))
    with open(path, "w") as f:
        f.write("".join(recs))
    limit, max_kmers = int(rng.integers(6000, 40000)), int(rng.integers(1500, 9000))
    # the C++ driver
    exe = hipbuild.build_host()
    out, prof = str(tmp_path / "out.jsonl"), str(tmp_path / "prof.csv")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", path, "-q", out, "-p", prof, "--jsonl", "-b", "100", "-n", "1", "-v"]
    cmd += ["--six"] if frames == 6 else (["--one"] if frames == 1 else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, KASA_LONG_SEQUENCE="1", KASA_PIECE_BYTES=str(limit), KASA_MAX_BATCH_KMERS=str(max_kmers)))
    assert r.returncode == 0, r.stderr[-2000:]
    sizes = [int(l.split()[3]) for l in r.stdout.splitlines() if l.startswith("OUT: Batch of")]
    # the same batches for the other two: the driver takes pieces while (letters + 64) x strands of them fit its device batch
    batch = reads.parse_reads(path)
    pieced = batch.with_pieces(12, frames, False, limit)
    if pieced is batch:
        pytest.skip("no record in more than one piece with this seed")
    strands = 2 if frames == 6 else 1
    bounds, est = [0], 0
    for q, tl in enumerate(np.diff(pieced.offsets)):
        k = (int(tl) + 64) * strands
        if q > bounds[-1] and est + k > max_kmers:
            bounds.append(q)
            est = 0
        est += k
    bounds.append(len(pieced.seg_read))
    seg = pieced.seg_read.astype(np.int64)
    assert sizes == [int(seg[b - 1] - seg[a]) + 1 for a, b in zip(bounds[:-1], bounds[1:])], r.stdout
    if not any(seg[b] == seg[b - 1] for b in bounds[1:-1]):
        pytest.skip("no batch ends inside a read with this seed")
    want_text, want_prof = _oracle_over_pieces(ix, pieced, bounds, frames)
    with open(out, "rb") as f:
        assert f.read().decode("latin-1") == want_text
    with open(prof, "rb") as f:
        assert f.read().decode("latin-1") == want_prof
    # the Python host
    run = identify.Identify(ix, 0, 12, 7, frames, 0.0, 100, "jsonl")
    run.piece_bytes, run.piece_bounds = limit, bounds
    text, ptext, _ = run.run(batch, True)
    assert run.batch_sizes == sizes
    assert text == want_text and ptext == want_prof
    run.close()


@pytest.mark.gpu
def test_cpp_host_pieces_tsv_and_filter(case, tmp_path):
    """The first input through the TSV writer (-b 3) with --filter: the reference's files (out_long_flt.tsv.gz; the filter's two
    files by their SHA-256)."""
    import hashlib
    from kasa_amd import build as hipbuild
    d, ix, data, gold = case
    exe = hipbuild.build_host()
    out, prof, c, x = (str(tmp_path / n) for n in ("out.tsv", "prof.csv", "c", "x"))
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "long.fasta"), "-q", out, "-p", prof,
           "--tsv", "-b", "3", "-m", "1", "-n", "1", "-v", "--filter", c, x]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [int(l.split()[3]) for l in r.stdout.splitlines() if l.startswith("OUT: Batch of")] == gold["long"]["batches"]
    with gzip.open(os.path.join(SRC, "out_long_flt.tsv.gz"), "rb") as f, open(out, "rb") as g:
        assert g.read() == f.read()
    with open(prof, "rb") as f, open(os.path.join(SRC, "prof_long.csv"), "rb") as g:
        assert f.read() == g.read()
    for mine, ref in ((c + ".fasta", "lflt_clean.fasta"), (x + ".fasta", "lflt_cont.fasta")):
        with open(mine, "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == gold["long_flt"]["sha256"][ref], ref
