"""GPU parity: the HIP path through the C ABI against (a) the reference binary's own output files,
(b) the CPU oracle on seeded inputs, stage by stage.  Bit-exact for k-mers, positions, unique counts
and float32 per-read scores; countAll (a double in the reference, an exact 64.64 fixed-point sum here)
within 1e-12 relative, and identical once printed with the reference's 6 significant digits."""
import os

import numpy as np
import pytest

from kasa_amd import capi, formats, reads, report
from kasa_amd.identify import Identify
from oracle import oracle
from tests import helpers
from tests.test_oracle_golden import PAIRS, _read, unpack, wants_coverage
from tests.test_oracle_properties import random_case

pytestmark = pytest.mark.gpu


def _gpu_or_fail():
    assert capi.device_count() > 0, "no HIP device visible: the GPU parity tests need a real MI355X"


def csr_rows(off, tax, sc):
    return [(tax[int(off[r]):int(off[r + 1])], sc[int(off[r]):int(off[r + 1])]) for r in range(off.shape[0] - 1)]


def assert_csr_equal(rows_gpu, rows_oracle):
    assert len(rows_gpu) == len(rows_oracle)
    for r, ((tg, sg), (to, so)) in enumerate(zip(rows_gpu, rows_oracle)):
        assert np.array_equal(tg, to), f"read {r}: taxa differ {tg} vs {to}"
        assert np.array_equal(sg.view(np.uint32), so.view(np.uint32)), f"read {r}: scores differ {sg} vs {so}"


@pytest.mark.parametrize("case", PAIRS, ids=[c[0] for c in PAIRS])
def test_golden_files_byte_identical(case):
    """End to end against the files the reference binary wrote."""
    _gpu_or_fail()
    stem, infile, fmt, kh, kl, frames, thr, beasts, idx, uniq = unpack(case)
    d, ix = helpers.load_case("pairs", idx)
    batch = reads.parse_reads(os.path.join(d, infile))
    idf = Identify(ix, 0, kh, kl, frames, thr, beasts, fmt, unique=uniq)
    text, prof, _ = idf.run(batch, coverage=wants_coverage(case))
    assert text == _read(os.path.join(d, "out_" + stem))
    assert prof == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))
    idf.close()


@pytest.mark.parametrize("frames", [3, 6, 1])
@pytest.mark.parametrize("krange", [(12, 7), (12, 12), (9, 6), (12, 5)])
def test_stages_vs_oracle_golden_inputs(frames, krange):
    """encode, sort, lookup depth, profile tables and per-read CSR against the oracle."""
    _gpu_or_fail()
    kh, kl = krange
    for name in ("pairs", "clones"):
        d, ix = helpers.load_case(name)
        batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
        p = oracle.params(kh, kl, frames)
        dix = capi.DeviceIndex(ix)
        ctx = capi.Context(dix, kh, kl, frames)
        ctx.upload(batch.bases, batch.offsets)
        n = ctx.encode()
        km_o, rd_o = oracle.encode(batch.bases, batch.offsets, p)
        km_g, rd_g = ctx.queries()
        assert n == km_o.shape[0]
        assert np.array_equal(km_g, km_o) and np.array_equal(rd_g, rd_o)
        ctx.sort_and_range()
        km_s, _ = ctx.queries()
        assert np.array_equal(km_s, np.sort(km_o))
        ctx.lookup_score(True)
        res, _ = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
        ca, cu, _ = ctx.profile()
        assert np.array_equal(cu, res.count_unique)
        np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
        assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
        ctx.close(); dix.close()


@pytest.mark.parametrize("slow", [0, 1, 2, 4, 1 | 16384 | 8388608, 134217728, 134217728 | 1, 1 | 1073741824, 1073741824],
                         ids=["fast+fallback", "general_score", "per_query_lookup", "sorting_row_merge", "general_third_pass", "cells_of_64_bytes", "cells_of_64_bytes_general",
                              "sorted_event_replay", "sorted_event_replay_of_the_fallbacks"])
@pytest.mark.parametrize("seed", range(24))
def test_adversarial_queries_vs_oracle(seed, slow):
    """Tiny alphabets, many taxa per k-mer, duplicates, '^' letters; queries cross several tiles.  (general_third_pass: every
    read through the general kernel's pending window in device memory.)"""
    _gpu_or_fail()
    rng = np.random.default_rng(5000 + seed)
    letters = [[1, 2], [1, 2, 30], [3, 4, 5, 30, 31], list(range(1, 21))][seed % 4]
    k_low = int(rng.integers(5, 12))
    k_high = int(rng.integers(k_low, 13))
    n_taxa = int(rng.integers(2, 40))
    n_reads = int(rng.integers(1, 50))
    ix, q, rd = random_case(seed, int(rng.integers(1, 3000)), int(rng.integers(5, 6000)), n_taxa, letters,
                            k_high, k_low, n_reads)
    p = oracle.params(k_high, k_low, 3)
    iv = oracle.IndexView(ix)
    qs, rs_ = oracle.sort_queries(q, rd)
    a, b = oracle.ranges(iv, p, qs)
    res = oracle.compare(iv, p, qs, rs_, a, b, n_reads, True, closed_form=False)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, k_high, k_low, 3)
    ctx.debug_flags(slow)
    ctx.set_queries(q, rd, n_reads)
    ctx.sort_and_range()
    ctx.lookup_score(True, coverage=False)
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    ctx.close(); dix.close()


def synth_genomes_of(seed, n_taxa, genome_len):
    """The genomes synthetic_world(seed, n_taxa, genome_len, ...) builds its index from."""
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = []
    for g in range(n_taxa):
        if g % 2 == 1:
            s = genomes[g - 1].copy()
            m = rng.random(genome_len) < 0.03
            s[m] = alphabet[rng.integers(0, 4, size=int(m.sum()))]
        else:
            s = alphabet[rng.integers(0, 4, size=genome_len)]
        genomes.append(s)
    return genomes


def synthetic_world(seed, n_taxa, genome_len, n_reads, read_len=150, K=12):
    rng = np.random.default_rng(seed)
    genomes = synth_genomes_of(seed, n_taxa, genome_len)
    content = formats.Content(["non_unique"] + [f"Taxon {g}" for g in range(n_taxa)],
                              np.concatenate(([0], 100 + np.arange(n_taxa))).astype(np.uint32))
    p = oracle.params(K, 7, 3, K=K)
    kms, tids = [], []
    for g, s in enumerate(genomes):  # index = forward k-mers of every genome (3 frames), as `build --three`
        km, _ = oracle.encode(s, np.array([0, genome_len], dtype=np.int64), p)
        kms.append(km)
        tids.append(np.full(km.shape[0], 100 + g, dtype=np.uint32))
    ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    batch = reads.synthetic_reads(genomes, n_reads, read_len, seed + 1)
    return ix, batch


def test_medium_synthetic_vs_oracle():
    """20 taxa x 20 kb, 4000 reads (520k queries, 500+ tiles): every stage against the oracle."""
    _gpu_or_fail()
    ix, batch = synthetic_world(7, 20, 20000, 4000)
    p = oracle.params(12, 7, 3)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    for slow in (0, 1, 2, 4, 32, 64):                           # 32: score_other_kernel per lane instead of flattened; 64: library sort over all bits
        ctx.debug_flags(slow)
        ctx.profile_reset()
        ctx.run_batch(batch.bases, batch.offsets, True)
        assert ctx.n_kmers == nq
        ca, cu, _ = ctx.profile()
        assert np.array_equal(cu, res.count_unique)
        np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
        assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
        if slow != 1:
            assert ctx.last_slow_reads() < batch.n // 10      # the lane-per-read path carries the load
    ctx.debug_flags(0)
    # profile-only mode gives the same tables; batches accumulate
    ctx.profile_reset()
    half = batch.n // 2
    for part in (batch.slice(0, half), batch.slice(half, batch.n)):
        ctx.run_batch(part.bases, part.offsets, False)
    ca2, cu2, _ = ctx.profile()
    assert np.array_equal(cu2, cu)
    np.testing.assert_allclose(ca2, ca, rtol=1e-12, atol=0)
    ctx.close(); dix.close()


def test_empty_and_degenerate_batches():
    _gpu_or_fail()
    d, ix = helpers.load_case("pairs")
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(np.zeros(0, np.uint8), np.zeros(1, np.int64), True)   # no reads
    off, tax, sc = ctx.scores()
    assert off.tolist() == [0] and tax.size == 0
    ctx.run_batch(np.frombuffer(b"ACGTACGTAC", dtype=np.uint8), np.array([0, 10], np.int64), True)  # 0 k-mers
    off, tax, sc = ctx.scores()
    assert off.tolist() == [0, 0]
    ctx.run_batch(np.frombuffer(b"ACGT", dtype=np.uint8), np.array([0, 0, 4, 4], np.int64), True)   # empty reads
    assert ctx.scores()[0].tolist() == [0, 0, 0, 0]
    with pytest.raises(RuntimeError):
        capi.Context(dix, 13, 7, 3)
    ctx2 = capi.Context(dix, 12, 7, 3)
    with pytest.raises(RuntimeError):
        ctx2.sort_and_range()                        # nothing encoded yet
    with pytest.raises(RuntimeError):
        ctx2.scores()
    ctx2.close(); ctx.close(); dix.close()


def test_coverage_counts_groups():
    """--coverage: countTotal counts each matched (level, prefix) group once (Compare.hpp:690)."""
    _gpu_or_fail()
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    p = oracle.params(12, 7, 3, coverage=True)
    iv = oracle.IndexView(ix)
    km, rd = oracle.sort_queries(*oracle.encode(batch.bases, batch.offsets, p))
    rs, rl = oracle.ranges(iv, p, km)
    res = oracle.compare(iv, p, km, rd, rs, rl, batch.n, True)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(batch.bases, batch.offsets, True, coverage=True)
    _, _, ct = ctx.profile()
    assert np.array_equal(ct, res.count_total)
    ctx.close(); dix.close()


def test_resident_upload_and_profile_only_mode():
    """kasa_batch_upload_device (bases and offsets in device memory, read in place: bench.py's step at every N) gives the batch
    kasa_batch_upload gives; and a batch without per-read output -- the profile is complete after the group stage, no score
    stage runs -- has the oracle's profile."""
    _gpu_or_fail()
    import torch
    ix, batch = synthetic_world(5, 12, 9000, 4000)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(batch.bases, batch.offsets, True)
    ref = (ctx.scores(), ctx.profile_limbs().copy())
    dev_bases = torch.from_numpy(np.ascontiguousarray(batch.bases)).to("cuda")
    dev_off = torch.from_numpy(np.ascontiguousarray(batch.offsets, dtype=np.int64)).to("cuda")
    torch.cuda.synchronize()
    for half in (False, True):                                   # the whole batch; then its second half (offsets not starting at 0)
        a = batch.n // 2 if half else 0
        ctx.profile_reset()
        ctx.upload_resident(dev_bases.data_ptr(), dev_off.data_ptr() + 8 * a, batch.n - a)
        ctx.encode(); ctx.sort_and_range(); ctx.lookup_score(True, False)
        if not half:
            got = (ctx.scores(), ctx.profile_limbs().copy())
            assert all(np.array_equal(x, y) for x, y in zip(got[0], ref[0])) and np.array_equal(got[1], ref[1])
        else:
            part = batch.slice(a, batch.n)
            res, nq = oracle.identify_batch(ix, part.bases, part.offsets, oracle.params(12, 7, 3), True)
            assert ctx.n_kmers == nq
            assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    # profile only
    res, _ = oracle.identify_batch(ix, batch.bases, batch.offsets, oracle.params(12, 7, 3), True)
    ctx.profile_reset()
    ctx.run_batch(batch.bases, batch.offsets, False)
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    assert ctx.stage_ms()["score"][1] >= 0
    with pytest.raises(RuntimeError):
        ctx.scores()                                             # there are none
    ctx.close(); dix.close()


def test_profile_limbs_roundtrip_and_sum():
    _gpu_or_fail()
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(batch.bases, batch.offsets, False)
    ca, cu, _ = ctx.profile()
    limbs = ctx.profile_limbs()
    ctx.profile_set_limbs(limbs * np.uint64(2))      # "two ranks with the same shard"
    ca2, cu2, _ = ctx.profile()
    assert np.array_equal(cu2, cu * np.uint64(2))
    np.testing.assert_allclose(ca2, 2 * ca, rtol=1e-15)
    ctx.close(); dix.close()


def _check_against_oracle(ix, batch, kh, kl, frames, flags=0, unique=False, protein=False):
    p = oracle.params(kh, kl, frames, K=ix.K, protein=protein)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True, unique=unique, seg_read=batch.seg_read,
                                    n_reads=batch.n)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, kl, frames)
    ctx.set_protein(protein)
    ctx.debug_flags(flags | int(os.environ.get("KASA_TEST_EXTRA_FLAGS", "0")))   # (bisecting a fuzz failure: one kernel choice against another)
    ctx.run_batch(batch.bases, batch.offsets, True, unique=unique, seg_read=batch.seg_read, n_reads=batch.n)
    assert ctx.n_kmers == nq
    if unique or protein:   # the stage outputs as well
        km_o, rd_o = oracle.encode(batch.bases, batch.offsets, p)
        if batch.seg_read is not None:
            rd_o = np.ascontiguousarray(batch.seg_read[rd_o])
        km_o, rd_o = oracle.sort_queries(km_o, rd_o)
        if unique:
            km_o, rd_o = oracle.unique_queries(km_o, rd_o)
        km_g, rd_g = ctx.queries()
        assert np.array_equal(km_g, km_o) and np.array_equal(rd_g, rd_o)
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    slow = ctx.last_slow_reads()
    ctx.close(); dix.close()
    return slow


@pytest.mark.parametrize("krange", [(12, 1), (12, 4), (8, 6), (11, 3)])
def test_wide_k_ranges(krange):
    """More than 6 levels and levels below the 6-letter prefix of the trie (general kernels only)."""
    _gpu_or_fail()
    ix, batch = synthetic_world(21, 8, 6000, 300)
    _check_against_oracle(ix, batch, krange[0], krange[1], 3)


def test_long_reads_and_mixed_lengths():
    """Reads of 1..6000 bases in one batch: encoder chunking, reads with zero k-mers, many k-mers per read."""
    _gpu_or_fail()
    rng = np.random.default_rng(5)
    ix, base = synthetic_world(33, 6, 8000, 10)
    gen = base.bases  # reuse random bases as a pool
    lens = [1, 2, 21, 22, 23, 35, 36, 37, 150, 151, 700, 1999, 6000, 0, 64, 513, 548, 1024]
    parts, off = [], [0]
    pool = np.concatenate([gen] * 8)
    for L in lens:
        a = int(rng.integers(0, pool.shape[0] - 6001))
        parts.append(pool[a:a + L])
        off.append(off[-1] + L)
    batch = reads.ReadBatch(np.concatenate(parts), np.asarray(off, dtype=np.int64), None,
                            np.asarray([l + 1 for l in lens], dtype=np.uint32))
    for frames in (3, 6, 1):
        _check_against_oracle(ix, batch, 12, 7, frames)
    _check_against_oracle(ix, batch, 12, 7, 3, unique=True)


@pytest.mark.parametrize("n_taxa", [24, 3000], ids=["slots_in_lds", "beyond_4096_taxa"])
def test_long_reads_keep_the_fast_kernels(n_taxa, monkeypatch):
    """Reads of thousands of k-mers with pieces of other taxa in them: their staging rows hold thousands of records (every query
    of a taxon that is not one of the read's two register taxa leaves up to a dozen).  Up to 4096 taxa the row merge streams such rows (its LDS arrays
    are per taxon of the row, not per record) and the reads stay on the lane-per-read kernels; round 5 handed every row beyond
    1024 records to score_dense_kernel -- what KASA_NO_LONG_ROWS=1 and indices beyond 4096 taxa still do.  Same bits either way."""
    _gpu_or_fail()
    rng = np.random.default_rng(91)
    g = synth_genomes_of(43, 24, 6000)
    ix, _ = synthetic_world(43, 24, 6000, 4)
    if n_taxa > 24:                                              # (the same k-mers under many more taxon ids: only the index's taxon count differs)
        content = formats.Content(["non_unique"] + [f"Taxon {t}" for t in range(n_taxa)], np.concatenate(([0], 100 + np.arange(n_taxa))).astype(np.uint32))
        p12 = oracle.params(12, 7, 3)
        kms, tids = [], []
        for t, sq in enumerate(g):
            km, _ = oracle.encode(sq, np.array([0, sq.shape[0]], dtype=np.int64), p12)
            kms.append(km); tids.append(np.full(km.shape[0], 100 + (t * 113) % n_taxa, dtype=np.uint32))
        ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    seqs = []
    for r in range(40):                                         # one stretch of 5000 bases + pieces of 260 bases of four other (unrelated) taxa
        t0 = int(rng.integers(0, 12)) * 2
        others = [t for t in rng.permutation(12) * 2 if t != t0][:4]
        a0 = int(rng.integers(0, 1000))
        seqs.append(np.concatenate([g[t0][a0:a0 + 5000]] + [g[t][a:a + 260] for t in others for a in [int(rng.integers(0, 5700))]]))
    seqs.append(g[3][:400].copy())
    off = np.concatenate(([0], np.cumsum([x.shape[0] for x in seqs]))).astype(np.int64)
    batch = reads.ReadBatch(np.concatenate(seqs), off, None, np.asarray([x.shape[0] + 1 for x in seqs], dtype=np.uint32))
    p = oracle.params(12, 7, 3)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    want = helpers.csr_from_dense(res.M)
    dix = capi.DeviceIndex(ix)
    for no_long in ("0", "1"):
        monkeypatch.setenv("KASA_NO_LONG_ROWS", no_long)
        ctx = capi.Context(dix, 12, 7, 3)
        ctx.run_batch(batch.bases, batch.offsets, True)
        st = ctx.batch_stats()
        general, _ = ctx.counters()
        if no_long == "0" and n_taxa <= 4096:
            # (a read whose FIRST deep matches in sorted order are a foreign piece's gets that piece's taxa as register taxa, and
            # its own 5000 queries fill the row beyond the cap; a read may also repeat a prefix of its own)
            assert st["dense_reads"] + general <= 12, (st, general)
        else:
            assert st["dense_reads"] + general >= 30, (st, general)
        ca, cu, _ = ctx.profile()
        assert np.array_equal(cu, res.count_unique)
        assert_csr_equal(csr_rows(*ctx.scores()), want)
        ctx.close()
    dix.close()


@pytest.mark.parametrize("frames,round_events", [(3, 0), (6, 0), (3, 600000), (1, 0)], ids=["three_frames", "six_frames", "several_rounds", "one_frame"])
def test_very_long_reads_replayed_from_sorted_events(frames, round_events, monkeypatch):
    """Reads of tens of thousands of k-mers (a contig: pieces of the genomes behind each other, regions met twice and three
    times, so that groups stay open across the read's later queries and one k-mer is hit several times) take the sorted-event
    replay (kasa_replay.h): events {read, taxon, flush position, level} sorted once, one float chain per (read, taxon) -- short
    chains a lane each, long ones a wavefront each.  Bit-equal to the oracle's sequential replay; with a budget of 600 000 events
    per round the two long reads go through several sorts, and a read beyond the budget goes back to the general kernel."""
    _gpu_or_fail()
    rng = np.random.default_rng(77 + frames)
    ix, base = synthetic_world(41, 8, 30000, 40)
    g = synth_genomes_of(41, 8, 30000)
    def contig(n_pieces):
        parts = []
        for _ in range(n_pieces):
            t = int(rng.integers(0, len(g)))
            a = int(rng.integers(0, 30000 - 4000))
            piece = g[t][a:a + int(rng.integers(500, 4000))].copy()
            parts.append(piece)
            if rng.integers(0, 3) == 0:
                parts.append(piece[: int(rng.integers(100, piece.shape[0]))].copy())      # the same region once more
        return np.concatenate(parts)
    long_a, long_b = contig(30), contig(12)
    seqs = [base.bases[base.offsets[i]:base.offsets[i + 1]] for i in range(5)] + [long_a] + \
           [base.bases[base.offsets[i]:base.offsets[i + 1]] for i in range(5, 12)] + [long_b, np.tile(g[0][:300], 70)]
    off = np.concatenate(([0], np.cumsum([x.shape[0] for x in seqs]))).astype(np.int64)
    batch = reads.ReadBatch(np.concatenate(seqs), off, None, np.asarray([x.shape[0] + 1 for x in seqs], dtype=np.uint32))
    if round_events:
        monkeypatch.setenv("KASA_ESR_ROUND_EVENTS", str(round_events))
    monkeypatch.setenv("KASA_ESR_MIN_KMERS", "16384" if frames != 1 else "5000")
    p = oracle.params(12, 7, frames)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, frames)
    ctx.run_batch(batch.bases, batch.offsets, True)
    st = ctx.batch_stats()
    assert st["replay_reads"] >= (2 if not round_events else 1) and st["replay_events"] > 0, st
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    ctx.debug_flags(536870912)                                    # (never the replay: the general kernel's answer is the same)
    ctx.profile_reset()
    ctx.run_batch(batch.bases, batch.offsets, True)
    assert ctx.batch_stats()["replay_reads"] == 0
    assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    ctx.close(); dix.close()


@pytest.mark.parametrize("krange,want_rows,K", [((25, 7), True, 25), ((25, 18), True, 25), ((25, 7), False, 25), ((12, 3), True, 12)],
                         ids=["k25_7", "k25_18", "profile_only", "64_byte_records_of_a_64_bit_index"])
def test_very_long_reads_against_a_128_bit_index(krange, want_rows, K, monkeypatch):
    """The same replay for 64-byte records (a 128-bit index, up to 19 levels): the sizes of a query's taxon sets are made from
    its segments, and -- the profile of such an index is not the group stage's -- every event also adds its exact share to the
    profile tables.  Rows and tables equal the oracle's; a profile-only run (kASA without -q) replays for the tables alone."""
    _gpu_or_fail()
    rng = np.random.default_rng(5)
    kh, kl = krange                                               # ((12, 3): ten levels -- 64-byte records with 64-bit keys; a fuzz seed found the replay reading them as narrow ones)
    ix, base = synthetic_world(47, 6, 30000, 30, K=K)
    g = synth_genomes_of(47, 6, 30000)
    parts = []
    for _ in range(14):
        t = int(rng.integers(0, len(g)))
        a = int(rng.integers(0, 30000 - 4000))
        piece = g[t][a:a + int(rng.integers(800, 4000))].copy()
        parts.append(piece)
        if rng.integers(0, 3) == 0:
            parts.append(piece[: int(rng.integers(100, piece.shape[0]))].copy())
    long_a = np.concatenate(parts)
    seqs = [base.bases[base.offsets[i]:base.offsets[i + 1]] for i in range(6)] + [long_a] + [base.bases[base.offsets[i]:base.offsets[i + 1]] for i in range(6, 10)] + [np.tile(g[1][:400], 50)]
    off = np.concatenate(([0], np.cumsum([x.shape[0] for x in seqs]))).astype(np.int64)
    batch = reads.ReadBatch(np.concatenate(seqs), off, None, np.asarray([x.shape[0] + 1 for x in seqs], dtype=np.uint32))
    p = oracle.params(kh, kl, 3, K=K)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, kl, 3)
    ctx.run_batch(batch.bases, batch.offsets, want_rows)
    st = ctx.batch_stats()
    if want_rows:                                                 # (without per-read scores no read breaks an order rule: the lane-per-read kernels keep them all)
        assert st["replay_reads"] >= 2 and st["replay_events"] > 0, st
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    if want_rows:
        assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
        ctx.debug_flags(536870912)                                # (never the replay)
        ctx.profile_reset()
        ctx.run_batch(batch.bases, batch.offsets, True)
        assert ctx.batch_stats()["replay_reads"] == 0
        assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
        ca2, cu2, _ = ctx.profile()
        assert np.array_equal(cu2, cu)
    ctx.close(); dix.close()


@pytest.mark.parametrize("frames", [3, 6])
@pytest.mark.parametrize("flags", [0, 1, 8])   # 8: slots from a sort by read id instead of the encoder's ranking
def test_unique_drops_repeats_inside_reads(frames, flags):
    """-e: tandem-repeat reads (the same k-mer many times in one read, also shared between reads) across several
    tiles; per-read k-mer counts change, so the per-read offsets are recomputed on the device."""
    _gpu_or_fail()
    rng = np.random.default_rng(77)
    ix, base = synthetic_world(41, 6, 3000, 400)
    pool = base.bases
    parts, off = [], [0]
    for r in range(300):
        a = int(rng.integers(0, pool.shape[0] - 200))
        unit = pool[a:a + int(rng.integers(20, 80))]
        seq = np.tile(unit, 8)[:int(rng.integers(40, 300))] if r % 3 else pool[a:a + 150]
        parts.append(seq)
        off.append(off[-1] + seq.shape[0])
    batch = reads.ReadBatch(np.concatenate(parts), np.asarray(off, dtype=np.int64), None,
                            np.asarray([p.shape[0] + 1 for p in parts], dtype=np.uint32))
    _check_against_oracle(ix, batch, 12, 7, frames, flags, unique=True)


@pytest.mark.parametrize("krange", [(12, 7), (12, 12), (10, 5)])
def test_protein_reads(krange):
    """Amino-acid input: letters 'A'..'Z', '*', lower case, lengths around K, reads longer than an encoder chunk."""
    _gpu_or_fail()
    rng = np.random.default_rng(88)
    ix, dna = synthetic_world(51, 6, 3000, 100)
    # translate stretches of the genomes the way the index was built, so that the reads match
    lut = oracle.codon_table()
    pool = dna.bases
    code = (pool & 14) >> 1
    parts, off = [], [0]
    for r, L in enumerate([0, 1, 5, 11, 12, 13, 14, 20, 50, 50, 64, 100, 333, 700, 30, 30]):
        a = int(rng.integers(0, pool.shape[0] - 3 * 701))
        c = code[a:a + 3 * L].reshape(-1, 3).astype(np.int64)
        aa = (lut[c[:, 0] * 64 + c[:, 1] * 8 + c[:, 2]] + 64).astype(np.uint8)
        aa[aa == ord("[")] = ord("*")
        if r % 4 == 1:
            aa = np.where((aa >= 65) & (aa <= 90), aa + 32, aa).astype(np.uint8)   # lower case
        parts.append(aa)
        off.append(off[-1] + L)
    batch = reads.ReadBatch(np.concatenate(parts), np.asarray(off, dtype=np.int64), None,
                            np.asarray([p.shape[0] + 1 for p in parts], dtype=np.uint32), True)
    slow = _check_against_oracle(ix, batch, krange[0], krange[1], 3, protein=True)
    _check_against_oracle(ix, batch, krange[0], krange[1], 6, protein=True)   # --six is ignored for protein input


def test_many_taxa_per_read_and_large_content():
    """A crowded index: 3000 taxa over 24 kb (every short prefix shared by many taxa) -> long staging rows, taxon
    sets of all sizes, the sorting row merge for > 16384 taxa is exercised through the debug flag as well."""
    _gpu_or_fail()
    rng = np.random.default_rng(9)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_taxa, L = 3000, 600
    root = alphabet[rng.integers(0, 4, size=L)]
    genomes = []
    for g in range(n_taxa):
        s = root.copy()
        m = rng.random(L) < 0.15
        s[m] = alphabet[rng.integers(0, 4, size=int(m.sum()))]
        genomes.append(s)
    content = formats.Content(["non_unique"] + [f"T{g}" for g in range(n_taxa)],
                              np.concatenate(([0], 10 + np.arange(n_taxa))).astype(np.uint32))
    p = oracle.params(12, 7, 3)
    kms, tids = [], []
    for g, s in enumerate(genomes):
        km, _ = oracle.encode(s, np.array([0, L], dtype=np.int64), p)
        kms.append(km)
        tids.append(np.full(km.shape[0], 10 + g, dtype=np.uint32))
    ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    batch = reads.synthetic_reads(genomes, 64, 150, 77)
    slow = _check_against_oracle(ix, batch, 12, 7, 3)
    _check_against_oracle(ix, batch, 12, 7, 3, flags=262144)     # long lists counted and placed by whole wavefronts (group_kernel's COOP form)
    _check_against_oracle(ix, batch, 12, 7, 3, flags=262144 | 1)
    _check_against_oracle(ix, batch, 12, 7, 3, flags=4)
    _check_against_oracle(ix, batch, 12, 7, 3, flags=32)
    _check_against_oracle(ix, batch, 12, 7, 3, flags=1)
    assert slow >= 0


@pytest.mark.parametrize("K", [12, 25])
def test_repeated_reads_fill_a_sort_bucket(K):
    """The query sort ranks the members of a bucket (equal top 32 key bits) by counting; a batch that repeats one read
    3000 times makes buckets of 3000 equal k-mers -- beyond SORT_BUCKET_LIMIT: they are sorted one by one (segmented
    radix sort), or, with debug flag 128, by the library over all bits (the last resort for too many / too long ones).  The
    sorted stream and everything after it must not notice (ties keep their batch order: the sort is stable)."""
    _gpu_or_fail()
    ix, batch = synthetic_world(61, 6, 5000, 700, K=K)
    one = batch.slice(3, 4)
    rep_b = np.concatenate([batch.bases] + [one.bases] * 3000)
    lens = [int(batch.offsets[r + 1] - batch.offsets[r]) for r in range(batch.n)] + [int(one.offsets[1])] * 3000
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    big = reads.ReadBatch(rep_b, off, [f"x{i} " for i in range(len(lens))], np.asarray([l + 1 for l in lens], dtype=np.uint32))
    kh = 12 if K == 12 else 25
    p = oracle.params(kh, 7, 3, K=K)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, 7, 3)
    outs = []
    for flags in (0, 64, 128, 512, 524288, 2097152, 2097152 | 524288, 2097152 | 128, 4194304):   # (524288: look-back first; 2097152: five passes, 20-bit buckets)
        ctx.debug_flags(flags)
        ctx.upload(big.bases, big.offsets); ctx.encode(); ctx.sort_and_range()
        km, rd = ctx.queries()
        outs.append((km.copy(), rd.copy()))
    km_o, rd_o = oracle.sort_queries(*oracle.encode(big.bases, big.offsets, p))
    for km, rd in outs:
        assert np.array_equal(km, km_o) and np.array_equal(rd, rd_o)
    ctx.close(); dix.close()
    _check_against_oracle(ix, big.slice(0, 1200), kh, 7, 3)


@pytest.mark.parametrize("tail", [200, 7, 1500])
@pytest.mark.parametrize("members", [500, 129, 1024])
def test_last_sort_bucket_straddles_the_last_tile(tail, members):
    """Round-2 advisor finding: the bucket pass of the query sort (bucket_rank_kernel; bucket_rank32_kernel with flag 2097152) scans a bucket left and right of
    every member inside a staged window; a bucket that begins more than the halo before the LAST tile and runs to the end of
    the batch must still be recognised as leaving the window.  The highest 8-letter prefix is repeated `members` times, the
    batch ends `tail` queries into its last 2048-query tile."""
    _gpu_or_fail()
    rng = np.random.default_rng(900 + tail + members)
    ix, _ = synthetic_world(71, 4, 3000, 10)
    n = 3 * 2048 + tail
    top = np.uint64(0x0F39CE739C) << np.uint64(20)                   # one 8-letter prefix, larger than everything else below
    q = rng.integers(1, 1 << 58, size=n, dtype=np.uint64)
    q[q >= top] >>= np.uint64(3)
    q[n - members:] = top | rng.integers(0, 1 << 20, size=members, dtype=np.uint64)
    rng.shuffle(q)
    rd = rng.integers(0, 50, size=n).astype(np.uint32)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    qs, rs_ = oracle.sort_queries(q, rd)
    for flags in (0, 64, 512, 524288, 2097152, 2097152 | 512, 4194304):   # (4194304: the bucket pass for any key width)
        ctx.debug_flags(flags)
        ctx.set_queries(q, rd, 50)
        ctx.sort_and_range()
        km, r2 = ctx.queries()
        assert np.array_equal(km, qs) and np.array_equal(r2, rs_)
    ctx.close(); dix.close()


@pytest.mark.parametrize("K", [12, 25])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 3 * 8192 + 5, 70001])
def test_query_sort_of_any_size(n, K):
    """The hand-written radix passes (kasa_radix.h: tiles of 8192 / 4096 pairs, look-back over the tiles before) followed
    by the bucket pass, on batches that end anywhere in a tile: equal to a stable sort, payload included.  Half of the keys
    share their top 40 bits with another key; a fifth are equal.  (Flags: 512 = the library's passes, 524288 = look-back first,
    2097152 = five passes and 20-bit buckets.)"""
    _gpu_or_fail()
    rng = np.random.default_rng(1234 + n)
    ix, _ = synthetic_world(71 + K, 4, 3000, 10, K=K)
    bits = 5 * K
    top = rng.integers(0, 1 << 40, size=max(1, n // 2), dtype=np.uint64)
    pick = top[rng.integers(0, top.shape[0], size=n)]
    if K == 12:
        q = (pick << np.uint64(20)) | rng.integers(0, 1 << 20, size=n, dtype=np.uint64)
        q[rng.random(n) < 0.2] = q[0]
    else:
        q = np.zeros(n, dtype=formats.KEY128_DTYPE)
        q["hi"] = (pick << np.uint64(21)) | rng.integers(0, 1 << 21, size=n, dtype=np.uint64)     # key bits 64..124
        q["lo"] = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
        q[rng.random(n) < 0.2] = q[0]
    rd = rng.integers(0, 30, size=n).astype(np.uint32)
    qs, rs_ = oracle.sort_queries(q, rd)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, K, 7, 3)
    for flags in (0, 512, 524288, 2097152, 2097152 | 524288, 4194304):
        ctx.debug_flags(flags)
        ctx.set_queries(q, rd, 30)
        ctx.sort_and_range()
        km, r2 = ctx.queries()
        assert np.array_equal(km, qs) and np.array_equal(r2, rs_)
    ctx.close(); dix.close()
    assert bits in (60, 125)


def test_one_frame_reads_of_several_encoder_chunks():
    """Round-2 advisor finding: --one gives the encoder chunks of 170 windows while it ranks up to 512 k-mers of a read
    itself; reads of 600-1500 bases (171..490 k-mers, no longer read in the batch) span several chunks."""
    _gpu_or_fail()
    rng = np.random.default_rng(6)
    ix, base = synthetic_world(35, 6, 8000, 10)
    pool = np.concatenate([base.bases] * 8)
    lens = [600, 640, 777, 900, 1024, 1199, 1200, 1500, 1540, 601]
    parts, off = [], [0]
    for L in lens:
        a = int(rng.integers(0, pool.shape[0] - 1600))
        parts.append(pool[a:a + L])
        off.append(off[-1] + L)
    batch = reads.ReadBatch(np.concatenate(parts), np.asarray(off, dtype=np.int64), None,
                            np.asarray([l + 1 for l in lens], dtype=np.uint32))
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 1)
    ctx.upload(batch.bases, batch.offsets); ctx.encode()
    assert ctx.batch_stats()["encoder_ranked"] == 1               # the path under test
    ctx.close(); dix.close()
    _check_against_oracle(ix, batch, 12, 7, 1)
    _check_against_oracle(ix, batch, 12, 9, 1)


def test_more_than_2_20_taxa():
    """A content file with more than 1 048 576 entries: staging records keep the taxon in 20 bits, so such an index must
    take the general score kernel for every read (and still give the oracle's numbers)."""
    _gpu_or_fail()
    rng = np.random.default_rng(21)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_taxa = (1 << 20) + 5000
    hot = [7, 1 << 19, (1 << 20) - 1, 1 << 20, (1 << 20) + 4321]   # the taxa that own k-mers (dense index = position in the content)
    genomes = [alphabet[rng.integers(0, 4, size=1500)] for _ in hot]
    genomes[3] = genomes[2].copy(); genomes[3][::40] = alphabet[rng.integers(0, 4, size=len(genomes[3][::40]))]   # siblings across the 2^20 line
    content = formats.Content(["non_unique"] + [f"T{g}" for g in range(n_taxa)], np.arange(n_taxa + 1, dtype=np.uint32) + np.uint32(10))
    content.taxids[0] = 0
    p = oracle.params(12, 7, 3)
    kms, tids = [], []
    for t, s in zip(hot, genomes):
        km, _ = oracle.encode(s, np.array([0, s.shape[0]], dtype=np.int64), p)
        kms.append(km)
        tids.append(np.full(km.shape[0], content.taxids[t], dtype=np.uint32))
    ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    batch = reads.synthetic_reads(genomes, 24, 150, 5)
    slow = _check_against_oracle(ix, batch, 12, 7, 3)
    assert slow == batch.n                                       # nothing may take the 20-bit fast path


# ---- 128-bit index (k <= 25) ----------------------------------------------------------------------------------------
from tests.test_oracle_golden import WIDE  # noqa: E402


@pytest.mark.parametrize("case", WIDE, ids=[c[0] for c in WIDE])
def test_wide_index_golden_inputs(case):
    """The reference's own --kH 25 index: every stage against the 128-bit oracle, and the rendered files against the
    oracle's closed form (the parity target: the stock binary compares 128-bit k-mers through a 64-bit functor,
    tests/test_oracle_golden.py::test_wide_index_stock_binary_and_parity_target)."""
    _gpu_or_fail()
    stem, kh, kl, frames = case
    d, ix = helpers.load_case("pairs", "idx25")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    p = oracle.params(kh, kl, frames, K=25)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, kl, frames)
    ctx.upload(batch.bases, batch.offsets)
    n = ctx.encode()
    km_o, rd_o = oracle.encode(batch.bases, batch.offsets, p)
    km_g, rd_g = ctx.queries()
    assert n == km_o.shape[0] and np.array_equal(km_g, km_o) and np.array_equal(rd_g, rd_o)
    ctx.sort_and_range()
    km_s, rd_s = ctx.queries()
    km_os, rd_os = oracle.sort_queries(km_o, rd_o)
    assert np.array_equal(km_s, km_os) and np.array_equal(rd_s, rd_os)
    ctx.lookup_score(True)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True, closed_form=True)
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    ctx.close()
    idf = Identify(ix, 0, kh, kl, frames, 0.0, 100, "jsonl", dix=dix)
    text, prof, _ = idf.run(batch)
    want = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique, nq, "jsonl",
                          kh, kl, frames, 0.0, 100)
    assert (text, prof) == want
    idf.close(); dix.close()


@pytest.mark.parametrize("krange", [(25, 7), (25, 20), (12, 7), (18, 13), (25, 1)])
@pytest.mark.parametrize("flags", [0, 2])
def test_wide_index_synthetic(krange, flags):
    """8 taxa x 6 kb with 128-bit keys, 400 reads over several tiles: fast (<= 6 levels) and general score kernels,
    streaming and per-query lookup, -e, --six."""
    _gpu_or_fail()
    ix, batch = synthetic_world(61, 8, 6000, 400, K=25)
    assert ix.K == 25
    _check_against_oracle(ix, batch, krange[0], krange[1], 3, flags)
    if krange == (25, 20):
        _check_against_oracle(ix, batch, 25, 20, 6, flags)
        _check_against_oracle(ix, batch, 25, 20, 3, flags, unique=True)


# ---- BASELINE.json's full size (C2): properties that do not need the oracle to run 1.3e9 queries ---------------------
@pytest.mark.parametrize("K,workload", [(12, "pairs"), (25, "pairs"), (12, "crowded")], ids=["C2_64bit_k12_7", "C3_128bit_k25_7", "crowded_64bit_k12_7"])
def test_full_size_properties(K, workload):
    """BASELINE.json configs[1] (64-bit index, -k 12 7) and configs[2] (128-bit index, -k 25 7, 64-byte event records) at
    full size.  10 M x 150 bp reads against the 4.2e8-record index of bench.py (scaled down with KASA_TEST_FULL_READS /
    KASA_TEST_FULL_TAXA when the box is small):
      * determinism: two runs give identical bytes;
      * linearity: the integer profile limbs of the whole batch equal the sum over five read shards (what the multi-GPU
        reduction relies on);
      * conservation: per level, sum over taxa of countAll = number of sorted queries whose match reaches that level;
      * shard invariance of the per-read taxon sets (scores may move in the last float digit with the batch);
      * a random sample of reads against the CPU oracle on the same 5 GB index;
      * the first 300 000 reads as a batch of their own: every score bit-equal to the oracle's on the same index.
    "crowded": the same index size, but the taxa come in clades of 50-200 that share conserved genes (synth.genomes_crowded;
    bench.py's `tertiary`): a third of the reads meet tens to hundreds of taxa per k-mer (long taxon lists, reads on the
    general score kernel, groups with thousands of hits).  2 M reads, 100 000 of them bit for bit against the oracle."""
    _gpu_or_fail()
    from kasa_amd import synth
    crowded = workload == "crowded"
    n_reads = int(os.environ.get("KASA_TEST_FULL_READS", "2000000" if crowded else "10000000"))
    n_taxa = int(os.environ.get("KASA_TEST_FULL_TAXA", "1400"))
    g = synth.genomes_crowded(n_taxa, 300_000, seed=11) if crowded else synth.genomes(n_taxa, 300_000, seed=11)
    kh = 12 if K == 12 else 25
    nK = kh - 7 + 1
    ix = synth.index_from_genomes(g, K=K)
    assert ix.K == K
    batch = synth.reads_from_genomes(g, n_reads, 150, seed=1000)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, 7, 3)
    assert ctx.rec_words == (8 if K == 12 else 16)

    def run(b):
        ctx.run_batch(b.bases, b.offsets, True)
        return ctx.scores()

    ctx.profile_reset()
    off1, tax1, sc1 = run(batch)
    limbs1 = ctx.profile_limbs().copy()
    ca, cu, _ = ctx.profile()
    depth, _ = ctx.lookup()
    # conservation
    for lv in range(nK):
        k = kh - lv
        matched = int(np.count_nonzero(depth >= k))
        assert abs(float(ca[lv].sum()) - matched) <= 1e-9 * max(1, matched), (k, float(ca[lv].sum()), matched)
    assert int(cu.sum()) > 0
    # determinism
    ctx.profile_reset()
    off2, tax2, sc2 = run(batch)
    assert np.array_equal(off1, off2) and np.array_equal(tax1, tax2) and np.array_equal(sc1.view(np.uint32), sc2.view(np.uint32))
    assert np.array_equal(limbs1, ctx.profile_limbs())
    del off2, tax2, sc2
    # linearity + shard invariance
    ctx.profile_reset()
    shards = 5
    for s in range(shards):
        a, b = s * batch.n // shards, (s + 1) * batch.n // shards
        o, t, v = run(batch.slice(a, b))
        lo, hi = int(off1[a]), int(off1[b])
        assert np.array_equal(o, off1[a:b + 1] - off1[a])
        assert np.array_equal(t, tax1[lo:hi])
        np.testing.assert_allclose(v, sc1[lo:hi], rtol=2e-5, atol=0)
    # limbs are stored un-normalised (independent accumulators): compare the folded 128-bit values
    def fold(l):
        l = l.reshape(-1, 6).astype(object)
        return [(int(r[0]), int(r[1]), int(r[2]) + (int(r[3]) << 32) + (int(r[4]) << 64) + (int(r[5]) << 96)) for r in l[::997]], \
            l[:, :2].astype(np.uint64)
    f1, u1 = fold(limbs1)
    f2, u2 = fold(ctx.profile_limbs())
    assert np.array_equal(u1, u2) and f1 == f2
    # oracle sample
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(batch.n, size=min(1500, batch.n), replace=False))
    sub_b = np.concatenate([batch.bases[int(batch.offsets[r]):int(batch.offsets[r + 1])] for r in pick])
    sub_o = np.concatenate(([0], np.cumsum([int(batch.offsets[r + 1] - batch.offsets[r]) for r in pick]))).astype(np.int64)
    res, _ = oracle.identify_batch(ix, sub_b, sub_o, oracle.params(kh, 7, 3, K=K), True)
    for i, r in enumerate(pick):
        t = (np.flatnonzero(res.M[i, 1:] > 0) + 1).astype(np.uint32)
        lo, hi = int(off1[r]), int(off1[r + 1])
        assert np.array_equal(tax1[lo:hi], t), r
        np.testing.assert_allclose(sc1[lo:hi], res.M[i, t], rtol=2e-5, atol=0)
    del res
    # a whole batch, bit for bit: the first reads as a batch of their own on the device and in the oracle (per-read
    # float sums depend on the reads that share a batch, so only equal batches can be compared exactly)
    n_exact = min(batch.n, int(os.environ.get("KASA_TEST_EXACT_READS", "100000" if crowded else "300000")))
    part = batch.slice(0, n_exact)
    ctx.profile_reset()
    o, t, v = run(part)
    ca_g, cu_g, _ = ctx.profile()
    res, nq = oracle.identify_batch(ix, part.bases, part.offsets, oracle.params(kh, 7, 3, K=K), True)
    assert ctx.n_kmers == nq
    assert np.array_equal(cu_g, res.count_unique)
    np.testing.assert_allclose(ca_g, res.count_all, rtol=1e-12, atol=0)
    rows, cols = np.nonzero(res.M[:, 1:] > 0)                     # row-major: reads ascending, taxa ascending
    assert np.array_equal(o, np.concatenate(([0], np.cumsum(np.bincount(rows, minlength=part.n)))).astype(np.uint64))
    assert np.array_equal(t, (cols + 1).astype(np.uint32))
    assert np.array_equal(v.view(np.uint32), res.M[rows, cols + 1].astype(np.float32).view(np.uint32))   # every float, every bit
    if crowded:                                                   # the workload is what it claims to be
        st = ctx.batch_stats()
        assert st["general_reads"] + st["dense_reads"] > n_exact // 20, st   # (reads with long rows: score_dense_kernel's, a few the general kernel's)
    ctx.close(); dix.close()


@pytest.mark.parametrize("case", [("pair", 3), ("pair6", 6)], ids=["pair", "pair6"])
def test_paired_end_golden_files(case):
    """-1/-2: two sequences per read through kasa_batch_upload_segments; also in two batches of pairs."""
    _gpu_or_fail()
    stem, frames = case
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_pairs(os.path.join(d, "pair_1.fastq"), os.path.join(d, "pair_2.fastq"))
    idf = Identify(ix, 0, 12, 7, frames, 0.0, 100, "jsonl")
    text, prof, _ = idf.run(batch)
    assert text == _read(os.path.join(d, "out_" + stem + ".jsonl"))
    assert prof == _read(os.path.join(d, "prof_" + stem + ".csv"))
    _, prof2, _ = idf.run(batch, batch_reads=5)          # the profile does not depend on the batching
    assert prof2 == prof
    idf.close()
    _check_against_oracle(ix, batch, 12, 7, frames, unique=True)


def test_two_contexts_share_one_index_concurrently():
    """identify_multiple (main.cpp:1292-1326): several drivers run at the same time on one read-only index.  Two host
    threads, each with its own context (stream, buffers, profile tables), different batches, same DeviceIndex."""
    _gpu_or_fail()
    import threading
    ix, batch = synthetic_world(71, 10, 8000, 3000)
    halves = [batch.slice(0, 1400), batch.slice(1400, 3000)]
    dix = capi.DeviceIndex(ix)
    out = [None, None]
    err = []

    def work(i):
        try:
            ctx = capi.Context(dix, 12, 7, 3 if i == 0 else 6)
            for _ in range(3):                                   # a few rounds, so the two really overlap
                ctx.profile_reset()
                ctx.run_batch(halves[i].bases, halves[i].offsets, True)
            out[i] = (ctx.scores(), ctx.profile())
            ctx.close()
        except Exception as e:  # noqa: BLE001
            err.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for i in range(2):
        p = oracle.params(12, 7, 3 if i == 0 else 6)
        res, _ = oracle.identify_batch(ix, halves[i].bases, halves[i].offsets, p, True)
        (off, tax, sc), (ca, cu, _) = out[i]
        assert np.array_equal(cu, res.count_unique)
        np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
        assert_csr_equal(csr_rows(off, tax, sc), helpers.csr_from_dense(res.M))
    dix.close()


@pytest.mark.parametrize("stem,K", [("idx", 12), ("idx25", 25)])
def test_index_build_matches_reference_files(stem, K, tmp_path):
    """`build` with the device encoder + sort (kasa_amd/index_build.py): index, trie, frequency and info files
    byte-identical to what the reference's build mode wrote for the same FASTA + content file."""
    _gpu_or_fail()
    from kasa_amd import index_build
    d = os.path.join(helpers.GOLDEN, "pairs")
    ix = index_build.build_index(os.path.join(d, "db.fasta"), os.path.join(d, "content.txt"), K=K)
    out = str(tmp_path / "new")
    formats.write_index(ix, out, str(tmp_path / "content_copy.txt"))
    for suffix in ("", "_trie", "_trie.txt", "_info.txt", "_f.txt"):
        assert _read(out + suffix) == _read(os.path.join(d, stem + suffix)), suffix


@pytest.mark.parametrize("case", [(12, 7, 3, 12, 0), (12, 7, 6, 12, 0), (25, 7, 3, 25, 0), (12, 1, 3, 12, 0), (12, 7, 3, 12, 1), (12, 7, 3, 12, 8),
                                  (25, 7, 6, 25, 8), (12, 7, 3, 12, 16), (25, 7, 3, 25, 16), (12, 7, 3, 12, 32)])
def test_profile_only_equals_per_read_run(case):
    """Without -q (kasa_batch_lookup_score(wantPerRead = 0)) the fast kernel skips ordering and float sums; the profile
    tables must be the same integers as those of the full run, and equal the oracle's."""
    _gpu_or_fail()
    kh, kl, frames, K, flags = case
    ix, batch = synthetic_world(91, 8, 7000, 2000, K=K)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, kl, frames)
    ctx.debug_flags(flags)
    ctx.run_batch(batch.bases, batch.offsets, True)
    full = ctx.profile_limbs().copy()
    ctx.profile_reset()
    ctx.upload(batch.bases, batch.offsets); ctx.encode(); ctx.sort_and_range(); ctx.lookup_score(False)
    assert np.array_equal(full, ctx.profile_limbs())
    res, _ = oracle.identify_batch(ix, batch.bases, batch.offsets, oracle.params(kh, kl, frames, K=K), False)
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    ctx.close(); dix.close()


def test_custom_codon_table_build_and_identify(tmp_path):
    """-a gc.prt 2: the index built on the device with the custom table is the reference's file, identify reproduces the
    reference's output, and the host-side parser gives the same table as the oracle's."""
    _gpu_or_fail()
    from kasa_amd import index_build
    d = os.path.join(helpers.GOLDEN, "pairs")
    lut = capi.codon_table_from_gcprt(os.path.join(d, "gc.prt"), "2")
    assert np.array_equal(lut, oracle.codon_table_from_file(os.path.join(d, "gc.prt"), "2"))
    assert np.array_equal(capi.builtin_codon_table(), oracle.codon_table())
    ix = index_build.build_index(os.path.join(d, "db.fasta"), os.path.join(d, "content.txt"), codon_lut=lut)
    formats.write_index(ix, str(tmp_path / "n"), str(tmp_path / "c.txt"))
    assert _read(str(tmp_path / "n")) == _read(os.path.join(d, "idxa"))
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    idf = Identify(ix, 0, 12, 7, 3, 0.0, 100, "jsonl", codon_lut=lut)
    text, prof, _ = idf.run(batch)
    assert text == _read(os.path.join(d, "out_alpha.jsonl"))
    assert prof == _read(os.path.join(d, "prof_alpha.csv"))
    idf.close()


@pytest.mark.parametrize("seed", range(150))
def test_random_configurations(seed):
    """Differential test over the option space: key width, k range, frames, -e, paired-end, read lengths (incl. reads
    shorter than a k-mer and reads longer than an encoder chunk), number of taxa, forced fallback paths."""
    _gpu_or_fail()
    rng = np.random.default_rng(9000 + seed)
    K = 25 if seed % 4 == 3 else 12
    k_high = int(rng.integers(1, K + 1))
    k_low = int(rng.integers(1, k_high + 1))
    if seed % 3 == 0:
        k_high, k_low = min(K, 12), 7
    frames = int(rng.choice([1, 3, 6]))
    unique = bool(rng.integers(0, 2))
    flags = int(rng.choice([0, 0, 1, 2, 4, 8, 9, 16, 20, 32, 40, 64, 72, 134217728, 134217728 | 32, 1 | 1073741824, 1073741824]))   # (134217728: narrow records in whole 64-byte cells; 1073741824: the general kernel's reads replayed from sorted events)
    n_taxa = int(rng.integers(2, 24))
    ix, base = synthetic_world(int(rng.integers(1, 1 << 30)), n_taxa, int(rng.integers(600, 4000)), 400, K=K)
    pool = base.bases
    n_reads = int(rng.integers(1, 300))
    parts, off, seg = [], [0], []
    paired = bool(rng.integers(0, 3) == 0)
    for r in range(n_reads):
        for _ in range(2 if paired else 1):
            L = int(rng.choice([0, 1, 20, 35, 36, 37, 40, 76, 77, 100, 150, 151, 300, 700, 1700]))
            a = int(rng.integers(0, max(1, pool.shape[0] - L)))
            s = pool[a:a + L].copy()
            if L and rng.integers(0, 5) == 0:
                s[int(rng.integers(0, L))] = ord("N")
            parts.append(s)
            off.append(off[-1] + s.shape[0])
            seg.append(r)
    batch = reads.ReadBatch(np.concatenate(parts) if parts else np.zeros(0, np.uint8), np.asarray(off, dtype=np.int64), None,
                            np.ones(n_reads, dtype=np.uint32), False, np.asarray(seg, dtype=np.uint32) if paired else None)
    if seed % 5 == 4:
        flags |= 268435456                                          # (the exact LDS accumulation of the profile keys on sparse taxon sets as well)
    _check_against_oracle(ix, batch, k_high, k_low, frames, flags, unique=unique)


@pytest.mark.parametrize("flags", [1, 1 | 16384, 1 | 8192, 262144, 262144 | 1, 1 | 16384 | 8388608, 1 | 16384 | 8388608 | 8192, 0, 33554432, 16777216, 268435456, 67108864 | 262144, 134217728],
                         ids=["dense_rows", "second_pass", "lane_owned_cells", "coop_group", "coop_group_general", "third_pass", "third_pass_lane_owned_cells",
                              "product_path", "no_dense_fast_kernel", "older_group_kernel", "exact_tables_always", "coop_group_table_cells", "cells_of_64_bytes"])
def test_general_kernel_on_huge_taxon_sets(flags):
    """Every taxon a light mutation of one root: a query meets hundreds of taxa per level.  On the general score kernel: with
    the read's row in LDS and an event's taxa dealt out to the lanes (the product path for such reads), through its second
    pass (debug flag 16384 hands every read on: the pass with the full pending window must give the same result), and in the
    lane-owns-its-cells form that indices beyond 16 384 taxa take (flag 8192).  Flag 8388608 lets the second pass hand every
    read on as well: to the third, whose pending window lies in device memory (what a read takes that keeps more than 4096
    groups pending -- a limit, KASA_E_LIMIT, until round 4)."""
    _gpu_or_fail()
    rng = np.random.default_rng(19)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_taxa, L = 1200, 400
    root = alphabet[rng.integers(0, 4, size=L)]
    genomes = []
    for g in range(n_taxa):                                     # every taxon a light mutation of one root: huge taxon sets
        s = root.copy()
        m = rng.random(L) < 0.02
        s[m] = alphabet[rng.integers(0, 4, size=int(m.sum()))]
        genomes.append(s)
    content = formats.Content(["non_unique"] + [f"T{g}" for g in range(n_taxa)],
                              np.concatenate(([0], 100 + np.arange(n_taxa))).astype(np.uint32))
    p = oracle.params(12, 7, 3)
    kms, tids = [], []
    for g, s in enumerate(genomes):
        km, _ = oracle.encode(s, np.array([0, L], dtype=np.int64), p)
        kms.append(km); tids.append(np.full(km.shape[0], 100 + g, dtype=np.uint32))
    ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    batch = reads.synthetic_reads(genomes, 60, 150, 5)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.debug_flags(flags)                                       # everything on the general kernel
    ctx.run_batch(batch.bases, batch.offsets, True)
    general, second = ctx.counters()
    assert (general == batch.n or not flags & 1) and (second == batch.n if flags & 16384 else second == 0), (general, second)
    assert ctx.third_pass_reads() == (batch.n if flags & 8388608 else 0)
    st = ctx.batch_stats()
    if not flags & (1 | 33554432):       # the fast kernels run: the long rows are score_dense_kernel's (these reads keep the order rule or go on)
        assert st["dense_reads"] > 0 and st["dense_reads"] + general == batch.n, (st, general)
    else:
        assert st["dense_reads"] == 0
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    ctx.close(); dix.close()


def test_record_buffer_is_chosen_among_candidates(monkeypatch):
    """The buffer of 32-byte event records is taken from candidates that are timed with scattered stores when it is allocated
    (reserve_placed: the rate of such stores is a property of the physical memory behind a buffer, tools/place_probe.hip).
    Buffers of 8 GB and more by default; KASA_PLACE_MIN_MB=0 sends this small batch through it: same batch bit for bit, the
    context says how it chose; KASA_PLACE_TRIES=1 and 64-byte records allocate plainly."""
    _gpu_or_fail()
    ix, batch = synthetic_world(5, 12, 9000, 4000)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(batch.bases, batch.offsets, True)
    ref = (ctx.scores(), ctx.profile_limbs().copy())
    assert ctx.record_placement()["candidates"] == 0                  # (a small buffer: plain)
    ctx.close()
    monkeypatch.setenv("KASA_PLACE_MIN_MB", "0")
    monkeypatch.setenv("KASA_PLACE_VERBOSE", "1")
    for tries, k_high in (("3", 12), ("4", 12), ("1", 12), ("3", 10)):
        monkeypatch.setenv("KASA_PLACE_TRIES", tries)
        ctx = capi.Context(dix, k_high, 7, 3)
        ctx.run_batch(batch.bases, batch.offsets, True)
        pl = ctx.record_placement()
        if tries == "1":
            assert pl["candidates"] == 0
        else:
            assert 1 <= pl["candidates"] <= int(tries) and pl["kept_g_records_per_s"] == max(pl["candidates_g_records_per_s"]) > 0, pl
        if k_high == 12:
            got = (ctx.scores(), ctx.profile_limbs().copy())
            assert all(np.array_equal(x, y) for x, y in zip(got[0], ref[0])) and np.array_equal(got[1], ref[1])
        ctx.run_batch(batch.bases, batch.offsets, True)               # (the buffer is there: nothing is chosen again)
        assert ctx.record_placement() == pl
        ctx.close()
    dix.close()
    ix16, batch16 = synthetic_world(5, 12, 9000, 2000, K=25)                 # 64-byte records are written as whole cells: plain
    dix = capi.DeviceIndex(ix16)
    ctx = capi.Context(dix, 25, 7, 3)
    ctx.run_batch(batch16.bases, batch16.offsets, True)
    assert ctx.record_placement()["candidates"] == 0
    ctx.close(); dix.close()


def test_tiles_with_heavy_groups_get_a_second_chance(monkeypatch):
    """A clade of 16 near-identical taxa among 30 unrelated ones: a query of the clade meets sixteen segments, and the tiles that
    hold many such leaders park more than group2_kernel's 1024 slots take but fewer than the 4096 of its second launch -- those
    tiles are grouped by that launch, not by the cooperative kernel, and the batch is bit-equal to the oracle either way
    (KASA_NO_SECOND_CHANCE=1: round 5's route).  (The world is dense on purpose: a tile's queries must not span more index entries
    than the kernel stages.)"""
    _gpu_or_fail()
    rng = np.random.default_rng(23)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = []
    root = alphabet[rng.integers(0, 4, size=10000)]
    for _ in range(16):
        s_ = root.copy()
        m = rng.random(10000) < 0.003
        s_[m] = alphabet[rng.integers(0, 4, size=int(m.sum()))]
        genomes.append(s_)
    for _ in range(30):
        genomes.append(alphabet[rng.integers(0, 4, size=10000)])
    n_taxa = len(genomes)
    content = formats.Content(["non_unique"] + [f"T{g}" for g in range(n_taxa)], np.concatenate(([0], 100 + np.arange(n_taxa))).astype(np.uint32))
    p = oracle.params(12, 7, 3)
    kms, tids = [], []
    for g, s_ in enumerate(genomes):
        km, _ = oracle.encode(s_, np.array([0, 10000], dtype=np.int64), p)
        kms.append(km); tids.append(np.full(km.shape[0], 100 + g, dtype=np.uint32))
    ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    batch = reads.synthetic_reads(genomes, 6000, 150, 9)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    want = helpers.csr_from_dense(res.M)
    dix = capi.DeviceIndex(ix)
    seen = {}
    for off in ("0", "1"):
        if off == "1":
            monkeypatch.setenv("KASA_NO_SECOND_CHANCE", "1")
        ctx = capi.Context(dix, 12, 7, 3)
        ctx.run_batch(batch.bases, batch.offsets, True)
        st = ctx.batch_stats()
        seen[off] = (st["group_tiles"], st["group_tiles_listed"], st["group_tiles_listed_again"])
        ca, cu, _ = ctx.profile()
        assert np.array_equal(cu, res.count_unique)
        np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
        assert_csr_equal(csr_rows(*ctx.scores()), want)
        ctx.close()
    dix.close()
    tiles, listed, again = seen["0"]
    assert 0 < listed <= tiles // 4 and again < listed, seen              # (the second launch kept listed tiles)
    assert seen["1"][2] == seen["1"][1] == listed, seen                    # (without it: everything listed is the cooperative kernel's)


def test_context_releases_every_device_buffer():
    """kasa_ctx_destroy gives back everything a context allocated, the third pass's window and list included (round 4
    leaked `gwin` and `ovList2`: up to 2 GB per context that ever ran the third pass)."""
    _gpu_or_fail()
    ix, batch = synthetic_world(5, 12, 6000, 400)
    dix = capi.DeviceIndex(ix)
    warm = capi.Context(dix, 12, 7, 3)                           # first use of the kernels: the runtime's own allocations
    warm.debug_flags(1 | 16384 | 8388608)
    warm.run_batch(batch.bases, batch.offsets, True)
    warm.close()
    free0, _ = capi.device_memory(0)
    for _ in range(3):
        ctx = capi.Context(dix, 12, 7, 3)
        ctx.debug_flags(1 | 16384 | 8388608)                     # every read through the second and the third pass
        ctx.run_batch(batch.bases, batch.offsets, True)
        assert ctx.third_pass_reads() == batch.n
        assert ctx.device_bytes() > 0
        ctx.close()
    free1, _ = capi.device_memory(0)
    assert free0 - free1 < (4 << 20), (free0, free1)             # (the allocator's granularity, not a buffer)
    dix.close()


def test_records_after_group_to():
    """kasa_batch_group_to writes the records into the caller's buffer; kasa_batch_records_fetch must hand out THOSE, not
    what an earlier batch left in the context's own buffer (advisor finding of round 4)."""
    _gpu_or_fail()
    import torch
    ix, batch = synthetic_world(7, 10, 8000, 900)
    dix = capi.DeviceIndex(ix)
    a = capi.Context(dix, 12, 7, 3)
    a.upload(batch.bases, batch.offsets); a.encode(); a.sort_and_range()
    ptr, n, kb = a.queries_device()
    b = capi.Context(dix, 12, 7, 3)
    half = n // 2
    b.set_sorted_device(ptr + half * kb, n - half)               # an earlier, different slice leaves its records in b's own buffer
    b.group()
    b.set_sorted_device(ptr, half)
    b.group()
    rec0, pool0 = b.records()
    out = torch.empty(half * b.rec_words, dtype=torch.int32, device="cuda:0")
    b.set_sorted_device(ptr + half * kb, n - half); b.group()     # ... and once more something else
    b.set_sorted_device(ptr, half)
    b.group_to(out.data_ptr())
    rec1, pool1 = b.records()
    def canon(rec, pool):                                       # pool blocks are handed out in the order the workgroups arrive
        rows = []
        for w in rec:
            n = int(w[3] & 255)
            if n <= 4:
                rows.append(tuple(int(x) for x in w))
                continue
            at = int(w[7])
            n = int(pool[at])
            skip = 4 if (int(w[2]) >> 30) & 1 else 0
            rows.append(tuple(int(x) for x in w[:7]) + tuple(int(x) for x in pool[at:at + 1 + skip + n - 3]))
        return rows
    assert canon(rec0, pool0) == canon(rec1, pool1)
    a.synchronize()
    assert canon(out.cpu().numpy().view(np.uint32).reshape(-1, b.rec_words), pool1) == canon(rec1, pool1)
    b.set_sorted_device(ptr, half)                               # a new batch: the caller's buffer is forgotten
    with pytest.raises(RuntimeError):
        b.records()
    a.close(); b.close(); dix.close()


@pytest.mark.parametrize("flags", [0], ids=["product_path"])
@pytest.mark.parametrize("n_taxa", [30, 300], ids=["lists_of_tens", "lists_of_hundreds"])
def test_wide_records_on_large_taxon_sets(n_taxa, flags):
    """64-byte records (128-bit index, -k 25 7) where a query meets tens or hundreds of taxa: every score equals the oracle's."""
    _gpu_or_fail()
    rng = np.random.default_rng(23)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    L = 600
    root = alphabet[rng.integers(0, 4, size=L)]
    genomes = []
    for g in range(n_taxa):
        s = root.copy()
        m = rng.random(L) < 0.03
        s[m] = alphabet[rng.integers(0, 4, size=int(m.sum()))]
        genomes.append(s)
    content = formats.Content(["non_unique"] + [f"T{g}" for g in range(n_taxa)],
                              np.concatenate(([0], 100 + np.arange(n_taxa))).astype(np.uint32))
    p = oracle.params(25, 7, 3, K=25)
    kms, tids = [], []
    for g, s in enumerate(genomes):
        km, _ = oracle.encode(s, np.array([0, L], dtype=np.int64), p)
        kms.append(km); tids.append(np.full(km.shape[0], 100 + g, dtype=np.uint32))
    ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    batch = reads.synthetic_reads(genomes, 80, 150, 9)
    res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, p, True)
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 25, 7, 3)
    ctx.debug_flags(flags)
    ctx.run_batch(batch.bases, batch.offsets, True)
    assert ctx.n_kmers == nq and ctx.rec_words == 16
    tiles, listed = ctx.group_tiles()
    assert listed == 0 or not flags, (tiles, listed)     # (the product path lists tiles for the older kernel -- or, once a pool retry has made the context sticky, runs it alone)
    ca, cu, _ = ctx.profile()
    assert np.array_equal(cu, res.count_unique)
    np.testing.assert_allclose(ca, res.count_all, rtol=1e-12, atol=0)
    assert_csr_equal(csr_rows(*ctx.scores()), helpers.csr_from_dense(res.M))
    ctx.close(); dix.close()
