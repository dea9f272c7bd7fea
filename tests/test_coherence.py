"""--coherence (Compare::postProcess, Compare.hpp:2607-2728).

CPU: the oracle's statement-by-statement restatement (ko_compare_sequential_ml + ko_coherence) rendered by the host
writers equals the files the reference binary wrote with --coherence (tests/golden/pairs/out_coh*, make_fixtures.py),
including the run it ends with an exception; the closed form's depth equals the match length setMatchLength leaves.
GPU: kasa_batch_coherence equals the oracle on inputs that exercise the walk's quirks, and both hosts write the
reference's bytes."""
import gzip
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from kasa_amd import capi, formats, reads, report
from oracle import oracle
from tests import helpers
from tests.test_oracle_golden import _read
from tests.test_oracle_properties import random_case

COH = [  # (output stem, input, fmt, kHigh, kLow, frames, beasts, index)
    ("coh.jsonl", "reads.fastq", "jsonl", 12, 7, 3, 100, "idx"),
    ("coh.tsv", "reads.fastq", "tsv", 12, 7, 3, 3, "idx"),
    ("coh.json", "reads.fastq", "json", 12, 7, 3, 3, "idx"),
    ("coh.ktsv", "reads.fastq", "kraken", 12, 7, 3, 3, "idx"),
    ("coh_fasta.tsv", "reads.fasta", "tsv", 12, 7, 3, 100, "idx"),
    ("coh_one.jsonl", "reads.fastq", "jsonl", 12, 7, 1, 100, "idx"),
    ("coh_k12_9.tsv", "reads.fastq", "tsv", 12, 9, 3, 100, "idx"),
    ("coh_prot.jsonl", "reads_prot.fasta", "jsonl", 12, 7, 3, 100, "idx"),
    ("coh_dup.tsv", "reads_dup.fastq", "tsv", 12, 7, 3, 100, "idx"),
    ("coh_dup6.tsv", "reads_dup.fastq", "tsv", 12, 7, 6, 100, "idx"),
]
FLAGS = {"json": "--json", "jsonl": "--jsonl", "tsv": "--tsv", "kraken": "--kraken"}


def _oracle_run(ix, batch, kh, kl, frames, closed_form=False, cmp64_quirk=False):
    p = oracle.params(kh, kl, frames, K=ix.K, protein=bool(batch.protein), cmp64_quirk=cmp64_quirk)
    return oracle.identify_batch_coherence(ix, batch.bases, batch.offsets, p, closed_form)


@pytest.mark.parametrize("case", COH, ids=[c[0] for c in COH])
@pytest.mark.parametrize("closed_form", [False, True], ids=["sequential", "closed_form"])
def test_oracle_coherence_files_byte_identical(case, closed_form):
    stem, infile, fmt, kh, kl, frames, beasts, idx = case
    d, ix = helpers.load_case("pairs", idx)
    batch = reads.parse_reads(os.path.join(d, infile))
    res, nq, coh, _ = _oracle_run(ix, batch, kh, kl, frames, closed_form)
    text, prof = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique, nq, fmt, kh, kl, frames,
                                0.0, beasts, protein=bool(batch.protein), coherence=coh)
    assert text == _read(os.path.join(d, "out_" + stem))
    assert prof == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))
    assert (coh != np.float32(12.0)).any() and (coh > 0).any()          # the case pins more than a constant


def test_oracle_coherence_wide_index():
    """128-bit index: the stock binary's 64-bit comparator quirk (cmp64Quirk) also decides its match lengths."""
    d, ix = helpers.load_case("pairs", "idx25")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    res, nq, coh, _ = _oracle_run(ix, batch, 25, 7, 3, cmp64_quirk=True)
    text, prof = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique, nq, "tsv", 25, 7, 3,
                                0.0, 100, coherence=coh)
    assert text == _read(os.path.join(d, "out_coh_w25_7.tsv"))
    assert prof == _read(os.path.join(d, "prof_coh_w25_7.csv"))


def test_oracle_reproduces_the_reference_exception():
    """--six --coherence on an input whose last strand switch finds no further match: the reference's walk asks its
    vector for the element behind the last one (vector::at, Compare.hpp:2667) and the run ends with that exception."""
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    with pytest.raises(oracle.ReferenceThrows) as e:
        _oracle_run(ix, batch, 12, 7, 6)
    assert "ERROR: " + str(e.value) + "\n" == _read(os.path.join(d, "coh_six_throws.err"))


def test_oracle_filter_with_coherence(tmp_path):
    """--filter: a read is a contaminant when its error is below --errorThreshold OR its coherence reaches
    --coherenceThreshold (Compare.hpp:1597-1606)."""
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    res, _, coh, _ = _oracle_run(ix, batch, 12, 7, 3)
    rows = helpers.csr_from_dense(res.M)
    flagged, by_error = [], 0
    for r in range(batch.n):
        rk = report.rank_read(rows[r][0], rows[r][1], int(batch.lengths[r]), ix.freq_at(12), 12, 7, 3, 0.0, 100)
        if not rk.hits:
            continue
        e = report.is_contaminant(rk.best, max(h.score for h in rk.hits), 0.46)
        by_error += int(e)
        if e or coh[r] >= np.float32(11.99):
            flagged.append(r)
    assert 0 < by_error < len(flagged) < batch.n                         # both rules decide something
    report.filter_reads([os.path.join(d, "reads.fastq")], flagged, str(tmp_path / "c"), str(tmp_path / "x"))
    assert _read(str(tmp_path / "c.fastq"), True) == _read(os.path.join(d, "cflt_clean.fastq"), True)
    assert _read(str(tmp_path / "x.fastq"), True) == _read(os.path.join(d, "cflt_cont.fastq"), True)


@pytest.mark.parametrize("seed", range(40))
def test_match_length_is_the_deepest_matched_level(seed):
    """What setMatchLength leaves in the sequential merge (last write wins, duplicates and stale level memories included)
    is the closed form's depth: the device looks it up per k-mer."""
    rng = np.random.default_rng(7000 + seed)
    letters = [[1, 2], [1, 2, 30], [3, 4, 5, 30, 31], list(range(1, 21))][seed % 4]
    k_low = int(rng.integers(5, 12)); k_high = int(rng.integers(k_low, 13))
    ix, q, rd = random_case(seed, int(rng.integers(1, 2000)), int(rng.integers(5, 4000)), int(rng.integers(2, 30)), letters,
                            k_high, k_low, 40)
    p = oracle.params(k_high, k_low, 3)
    iv = oracle.IndexView(ix)
    qs, rs_ = oracle.sort_queries(q, rd)
    a, b = oracle.ranges(iv, p, qs)
    r1, ml = oracle.compare_ml(iv, p, qs, rs_, a, b, 40, closed_form=False)
    r2, dp = oracle.compare_ml(iv, p, qs, rs_, a, b, 40, closed_form=True)
    assert np.array_equal(ml, dp)
    assert np.array_equal(r1.M.view(np.uint32), r2.M.view(np.uint32))


def _walk_reference(rd, pos, frame, ml, six, n_reads):
    """ko_coherence once more in plain Python (small inputs): guards the C restatement against typos."""
    sc = np.zeros(n_reads, dtype=np.float32)
    n = len(rd); idx = 0; rid = 0; last = 0; cur = 0; cnt = 0
    f32 = np.float32
    def cluster():
        with np.errstate(divide="ignore"):
            v = f32(f32(cur) + f32(1.0)) - f32(1.0) / f32(cnt)
        if sc[rid] < v: sc[rid] = v
    while idx < n:
        if ml[idx]:
            rid = int(rd[idx]); last = int(pos[idx]) + int(ml[idx]); idx += 1; break
        idx += 1
    while rid < n_reads and idx < n:
        fb = 0
        while fb < (2 if six else 1):
            if idx >= n: raise IndexError(idx)
            m = int(ml[idx])
            if m:
                ps = int(pos[idx])
                if ps <= last:
                    nx = m if ps + m < last else ((last - ps) & 0xFFFFFFFF)
                    if nx > cur: cur, cnt = nx, 1
                    elif nx == cur: cnt += 1
                else:
                    cluster(); cur = 0
                last = ps + m
            idx += 1
            if idx == n: cluster(); break
            if int(rd[idx]) != rid:
                cluster(); last = 0xFFFFFFFF; cur = 0; cnt = 0; break
            if int(frame[idx]) != fb:
                cluster(); cur = 0; cnt = 0; fb += 1
                while idx < n:
                    if ml[idx]:
                        last = int(pos[idx]) + int(ml[idx]); idx += 1; break
                    idx += 1
        rid += 1
    return sc


@pytest.mark.parametrize("seed", range(30))
def test_walk_restatement_on_random_streams(seed):
    rng = np.random.default_rng(8100 + seed)
    six = bool(seed % 2)
    n_reads = int(rng.integers(1, 30))
    rd, pos, fr = [], [], []
    for r in range(n_reads):
        c = int(rng.integers(0, 12)) if rng.random() < 0.8 else 0
        for s in range(2 if six else 1):
            rd += [r] * c; pos += list(range(c)); fr += [s] * c
    n = len(rd)
    dens = [0.0, 0.05, 0.3, 0.9][seed % 4]
    ml = (rng.integers(5, 13, size=n) * (rng.random(n) < dens)).astype(np.uint8)
    try:
        want = _walk_reference(rd, pos, fr, ml, six, n_reads)
    except IndexError:
        want = None
    try:
        got = oracle.coherence(np.asarray(rd, np.uint32), np.asarray(pos, np.uint32), np.asarray(fr, np.uint8), ml, six, n_reads)
    except oracle.ReferenceThrows:
        got = None
    assert (want is None) == (got is None)
    if want is not None:
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def _unpacked_batches(tmp_path):
    """tests/golden/batches with its gzip'ed parts unpacked (as tests/test_batches.py does)."""
    src = os.path.join(helpers.GOLDEN, "batches")
    d = str(tmp_path)
    for f in os.listdir(src):
        if f.endswith(".gz") and not f.startswith("reads"):
            with gzip.open(os.path.join(src, f), "rb") as g, open(os.path.join(d, f[:-3]), "wb") as o:
                shutil.copyfileobj(g, o)
        else:
            shutil.copy(os.path.join(src, f), os.path.join(d, f))
    ix = formats.load_index(os.path.join(d, "idx"), os.path.join(d, "content.txt"))
    batch = reads.parse_reads(os.path.join(d, "reads.fastq.gz"))
    return d, ix, batch, json.load(open(os.path.join(d, "batches.json")))


def test_refbatch_with_coherence_reproduces_the_reference_batch_sizes(tmp_path):
    """--coherence widens the reference's k-mer records (ppTuple) and adds a float per read: other batch boundaries."""
    d, ix, batch, sizes = _unpacked_batches(tmp_path)
    rb = capi.RefBatcher(ix, 12, 7, 3, memory_gib=1, threads=1, coherence=True)
    assert list(np.diff(rb.boundaries(batch, True))) == sizes["m1_coh"]
    assert sizes["m1_coh"] != sizes["m1"]


def test_oracle_coherence_batch_by_batch_equals_the_reference(tmp_path):
    d, ix, batch, sizes = _unpacked_batches(tmp_path)
    bounds = [0] + list(np.cumsum(sizes["m1_coh"]))
    rows, cohs, ca, cu, nq = [], [], None, None, 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        part = batch.slice(a, b)
        res, n, coh, _ = _oracle_run(ix, part, 12, 7, 3, True)
        for r in range(part.n):
            t = np.flatnonzero(res.M[r, 1:] > 0) + 1
            rows.append((t.astype(np.uint32), res.M[r, t].astype(np.float32)))
        cohs.append(coh)
        ca = res.count_all if ca is None else ca + res.count_all
        cu = res.count_unique if cu is None else cu + res.count_unique
        nq += n
        del res
    text, prof = helpers.render(ix, batch, rows, ca, cu, nq, "jsonl", 12, 7, 3, 0.0, 100, coherence=np.concatenate(cohs))
    assert text == _read(os.path.join(d, "out_m1_coh.jsonl"))
    assert prof == _read(os.path.join(d, "prof_m1_coh.csv"))


# ------------------------------------------------------------------------------------------------ GPU
def _device_coherence(ix, batch, kh, kl, frames):
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, kl, frames)
    ctx.set_protein(bool(batch.protein))
    ctx.run_batch(batch.bases, batch.offsets, True)
    try:
        return ctx.coherence()
    finally:
        ctx.close(); dix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", COH, ids=[c[0] for c in COH])
def test_device_coherence_equals_the_oracle_on_golden_inputs(case):
    stem, infile, fmt, kh, kl, frames, beasts, idx = case
    d, ix = helpers.load_case("pairs", idx)
    batch = reads.parse_reads(os.path.join(d, infile))
    _, _, coh, _ = _oracle_run(ix, batch, kh, kl, frames, True)
    got = _device_coherence(ix, batch, kh, kl, frames)
    assert np.array_equal(got.view(np.uint32), coh.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [3, 6, 1])
@pytest.mark.parametrize("seed", range(4))
def test_device_coherence_many_chunks_and_quirks(frames, seed):
    """Thousands of reads (many chunks of the parallel walk), reads without k-mers in runs (each takes an element of its
    successor), foreign reads (no match: with --six the search after the strand switch runs through them), reads
    matching on one strand only, and a last read matching on both strands so that the reference does not throw."""
    from tests.test_gpu_parity import synthetic_world
    rng = np.random.default_rng(9200 + seed)
    ix, base = synthetic_world(300 + seed, 6, 6000, 1500)
    comp = np.zeros(256, dtype=np.uint8); comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    parts = []
    for r in range(base.n):
        s = base.bases[base.offsets[r]:base.offsets[r + 1]]
        u = rng.random()
        if u < 0.08:
            s = s[:int(rng.integers(1, 22))]                            # no k-mers at all
        elif u < 0.20:
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=150)]   # foreign
        elif u < 0.30:
            s = s[:int(rng.integers(23, 60))]
        parts.append(s)
    last = base.bases[base.offsets[0]:base.offsets[1]][:75]
    parts.append(np.concatenate((last, comp[last[::-1]])))               # matches on both strands
    off = np.concatenate(([0], np.cumsum([len(x) for x in parts]))).astype(np.int64)
    batch = reads.ReadBatch(np.concatenate(parts), off, None, np.asarray([len(x) + 1 for x in parts], dtype=np.uint32))
    p = oracle.params(12, 7, frames)
    _, _, coh, _ = oracle.identify_batch_coherence(ix, batch.bases, batch.offsets, p, True)
    got = _device_coherence(ix, batch, 12, 7, frames)
    assert np.array_equal(got.view(np.uint32), coh.view(np.uint32))
    assert len(np.unique(coh)) > 5


@pytest.mark.gpu
@pytest.mark.parametrize("matches", [True, False], ids=["first_match_late", "no_match_at_all"])
def test_device_coherence_first_match_late_in_a_large_batch(matches):
    """Hundreds of thousands of reads that match nothing (thousands of chunks that take no turn) before the first matching read: the
    chunks without a turn leave where their successor starts, so the fix-up settles in a round or two (advisor finding of
    round 3: one round per chunk -- a launch, a copy and a synchronisation each)."""
    import time
    from tests.test_gpu_parity import synthetic_world
    rng = np.random.default_rng(77)
    ix, base = synthetic_world(411, 6, 6000, 400)
    n_foreign = 300_000
    foreign = np.full((n_foreign, 100), ord("N"), dtype=np.uint8)          # (random reads find chance matches at k = 7; these cannot)
    tail = base.bases[:base.offsets[base.n]] if matches else np.zeros(0, dtype=np.uint8)
    off = np.concatenate((np.arange(n_foreign + 1, dtype=np.int64) * 100,
                          (n_foreign * 100 + base.offsets[1:base.n + 1]) if matches else np.zeros(0, dtype=np.int64))).astype(np.int64)
    bases = np.concatenate((foreign.reshape(-1), tail))
    p = oracle.params(12, 7, 3)
    _, _, coh, _ = oracle.identify_batch_coherence(ix, bases, off, p, False)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(bases, off, True)
    ctx.synchronize()
    t0 = time.perf_counter()
    got = ctx.coherence()
    dt = time.perf_counter() - t0
    ctx.close(); dix.close()
    assert np.array_equal(got.view(np.uint32), coh.view(np.uint32))
    assert (coh > 0).any() == matches
    # (a guard against the round-per-chunk pathology -- minutes for this input -- not a timing test: a busy or freshly started
    # box may take a second or two for what normally takes 50 ms)
    assert dt < 30.0, f"{dt:.2f} s: the walk's fix-up rounds did not settle"


@pytest.mark.gpu
def test_device_reports_where_the_reference_throws():
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    with pytest.raises(RuntimeError) as e:
        _device_coherence(ix, batch, 12, 7, 6)
    assert "ERROR: " + str(e.value) + "\n" == _read(os.path.join(d, "coh_six_throws.err"))


@pytest.mark.gpu
def test_device_coherence_rejects_what_the_reference_leaves_to_chance():
    d, ix = helpers.load_case("pairs")
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    batch = reads.parse_reads(os.path.join(d, "reads_dup.fastq"))
    ctx.run_batch(batch.bases, batch.offsets, True, unique=True)
    with pytest.raises(RuntimeError, match="-e"):
        ctx.coherence()
    pairs = reads.parse_pairs(os.path.join(d, "pair_1.fastq"), os.path.join(d, "pair_2.fastq"))
    ctx.run_batch(pairs.bases, pairs.offsets, True, seg_read=pairs.seg_read, n_reads=pairs.n)
    with pytest.raises(RuntimeError, match="paired"):
        ctx.coherence()
    ctx.close(); dix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", COH, ids=[c[0] for c in COH])
def test_python_host_coherence_files_byte_identical(case):
    from kasa_amd.identify import Identify
    stem, infile, fmt, kh, kl, frames, beasts, idx = case
    d, ix = helpers.load_case("pairs", idx)
    batch = reads.parse_reads(os.path.join(d, infile))
    idf = Identify(ix, 0, kh, kl, frames, 0.0, beasts, fmt, coherence=True)
    text, prof, _ = idf.run(batch)
    assert text == _read(os.path.join(d, "out_" + stem))
    assert prof == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))
    idf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", COH, ids=[c[0] for c in COH])
def test_cpp_host_coherence_files_byte_identical(case, tmp_path):
    from kasa_amd import build as hipbuild
    exe = hipbuild.build_host()
    stem, infile, fmt, kh, kl, frames, beasts, idx = case
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, idx), "-i", os.path.join(d, infile),
           "-q", out, "-p", prof, FLAGS[fmt], "-b", str(beasts), "-k", str(kh), str(kl), "-m", "4", "-n", "1", "--coherence"]
    cmd += {6: ["--six"], 1: ["--one"]}.get(frames, [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert _read(out) == _read(os.path.join(d, "out_" + stem))
    assert _read(prof) == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))


@pytest.mark.gpu
def test_cpp_host_coherence_exception_and_filter(tmp_path):
    from kasa_amd import build as hipbuild
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    base = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq"), "-n", "1"]
    r = subprocess.run(base + ["--tsv", "--six", "--coherence", "-q", str(tmp_path / "o"), "-p", str(tmp_path / "p")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 1
    assert _read(os.path.join(d, "coh_six_throws.err")).strip() in r.stderr
    c, x = str(tmp_path / "c"), str(tmp_path / "x")
    r = subprocess.run(base + ["--jsonl", "-b", "100", "--coherence", "--coherenceThreshold", "11.99", "--errorThreshold", "0.46",
                               "--filter", c, x, "-p", str(tmp_path / "p2")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert _read(c + ".fastq", True) == _read(os.path.join(d, "cflt_clean.fastq"), True)
    assert _read(x + ".fastq", True) == _read(os.path.join(d, "cflt_cont.fastq"), True)


@pytest.mark.gpu
def test_hosts_coherence_across_the_reference_batches(tmp_path):
    """-m 1 --coherence: three batches with the wider records' boundaries; every batch is walked on its own."""
    from kasa_amd import build as hipbuild, identify
    d, ix, batch, sizes = _unpacked_batches(tmp_path)
    sizes = sizes["m1_coh"]
    run = identify.Identify(ix, 0, 12, 7, 3, 0.0, 100, "jsonl", coherence=True)
    text, prof, _ = run.run(batch, True, memory_gib=1, threads=1)
    assert run.batch_sizes == sizes
    assert text == _read(os.path.join(d, "out_m1_coh.jsonl"))
    assert prof == _read(os.path.join(d, "prof_m1_coh.csv"))
    run.close()
    exe = hipbuild.build_host()
    out, pf = str(tmp_path / "o.jsonl"), str(tmp_path / "p.csv")
    r = subprocess.run([exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq.gz"),
                        "-q", out, "-p", pf, "--jsonl", "-b", "100", "-m", "1", "-n", "1", "--coherence"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert _read(out) == _read(os.path.join(d, "out_m1_coh.jsonl"))
    assert _read(pf) == _read(os.path.join(d, "prof_m1_coh.csv"))
