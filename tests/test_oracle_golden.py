"""Pin the oracle: its results, rendered by the host-side report module, must equal byte for byte
what the reference's shipped binary wrote for the same inputs (tests/golden/, make_fixtures.py)."""
import os

import pytest

from kasa_amd import reads
from tests import helpers

PAIRS = [  # (output stem, input file, fmt, kHigh, kLow, frames, threshold, beasts[, index stem[, -e[, --coverage]]])
    ("default.json", "reads.fastq", "json", 12, 7, 3, 0.0, 3),
    ("b100.jsonl", "reads.fastq", "jsonl", 12, 7, 3, 0.0, 100),
    ("b100.tsv", "reads.fastq", "tsv", 12, 7, 3, 0.0, 100),
    ("b1.tsv", "reads.fastq", "tsv", 12, 7, 3, 0.0, 1),
    ("default.ktsv", "reads.fastq", "kraken", 12, 7, 3, 0.0, 3),
    ("fasta.jsonl", "reads.fasta", "jsonl", 12, 7, 3, 0.0, 100),
    ("k12_9.jsonl", "reads.fastq", "jsonl", 12, 9, 3, 0.0, 100),
    ("k10_7.jsonl", "reads.fastq", "jsonl", 10, 7, 3, 0.0, 100),
    ("k12_12.jsonl", "reads.fastq", "jsonl", 12, 12, 3, 0.0, 100),
    ("six.jsonl", "reads.fastq", "jsonl", 12, 7, 6, 0.0, 100),
    ("thr04.jsonl", "reads.fastq", "jsonl", 12, 7, 3, 0.4, 100),
    ("ram.jsonl", "reads.fastq", "jsonl", 12, 7, 3, 0.0, 100),
    ("exampleInput.jsonl", "exampleInput.fasta", "jsonl", 12, 7, 3, 0.0, 100),
    ("half.jsonl", "reads.fastq", "jsonl", 12, 7, 3, 0.0, 100, "idx_half"),   # 6-byte index of shrink strategy 2
    ("one.jsonl", "reads.fastq", "jsonl", 12, 7, 1, 0.0, 100),                # --one
    ("prot.jsonl", "reads_prot.fasta", "jsonl", 12, 7, 3, 0.0, 100),          # amino-acid input (detected from the file)
    ("dup.jsonl", "reads_dup.fastq", "jsonl", 12, 7, 3, 0.0, 100),            # reads repeating their own k-mers ...
    ("unique.jsonl", "reads_dup.fastq", "jsonl", 12, 7, 3, 0.0, 100, "idx", True),   # ... and the same with -e
    ("unique6.jsonl", "reads_dup.fastq", "jsonl", 12, 7, 6, 0.0, 100, "idx", True),  # -e --six
    ("edge_crlf.jsonl", "edge_crlf.fasta", "jsonl", 12, 7, 3, 0.0, 100),     # CRLF: the '\r' stays in name, length, k-mers
    ("edge_multi.jsonl", "edge_multi.fastq", "jsonl", 12, 7, 3, 0.0, 100),   # header with spaces, '+name', '#' qualities
    ("edge_noeol.jsonl", "edge_noeol.fasta", "jsonl", 12, 7, 3, 0.0, 100),   # no line feed at the end of the file
    ("cov.jsonl", "reads.fastq", "jsonl", 12, 7, 3, 0.0, 100, "idx", False, True),   # --coverage: two more column groups
]


def unpack(case):
    return case[:8] + ((case[8],) if len(case) > 8 else ("idx",)) + ((case[9],) if len(case) > 9 else (False,))


def wants_coverage(case):
    return len(case) > 10 and bool(case[10])


def _read(path, binary=False):
    with open(path, "rb") as f:
        data = f.read()
    return data if binary else data.decode("latin-1")


@pytest.mark.parametrize("case", PAIRS, ids=[c[0] for c in PAIRS])
@pytest.mark.parametrize("closed_form", [False, True], ids=["sequential", "closed_form"])
def test_pairs_byte_identical(case, closed_form):
    stem, infile, fmt, kh, kl, frames, thr, beasts, idx, uniq = unpack(case)
    d, ix = helpers.load_case("pairs", idx)
    batch = reads.parse_reads(os.path.join(d, infile))
    assert batch.protein == (stem == "prot.jsonl")
    cov = wants_coverage(case)
    res, nq = helpers.oracle_identify(ix, batch, kh, kl, frames, closed_form=closed_form, unique=uniq,
                                      protein=batch.protein, coverage=cov)
    text, prof = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique,
                                nq, fmt, kh, kl, frames, thr, beasts, protein=batch.protein,
                                count_total=res.count_total if cov else None)
    assert text == _read(os.path.join(d, "out_" + stem))
    assert prof == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))


@pytest.mark.parametrize("closed_form", [False, True], ids=["sequential", "closed_form"])
def test_clones_avx_build(closed_form):
    """> 3 taxa per k-mer: the shipped binary is an AVX build, the oracle's avxQuirk flag restates
    that branch (Compare.hpp:534-597).  The parity target for the device is avxQuirk = 0."""
    d, ix = helpers.load_case("clones")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    res, nq = helpers.oracle_identify(ix, batch, avx_quirk=True, closed_form=closed_form)
    text, prof = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique,
                                nq, "jsonl", 12, 7, 3, 0.0, 100)
    assert text == _read(os.path.join(d, "out_b100.jsonl"))
    assert prof == _read(os.path.join(d, "prof_b100.csv"))
    assert prof == _read(os.path.join(d, "prof_only.csv"))
    # and the scalar branch really differs on this data (otherwise the case pins nothing)
    res2, _ = helpers.oracle_identify(ix, batch, avx_quirk=False)
    assert (res2.M != res.M).any()


WIDE = [  # 128-bit index (build --kH 25): (output stem, kHigh, kLow, frames)
    ("w25_7", 25, 7, 3), ("w12_7", 12, 7, 3), ("w25_20", 25, 20, 3), ("w16_9", 16, 9, 3), ("w25_7_six", 25, 7, 6),
]


@pytest.mark.parametrize("case", WIDE, ids=[c[0] for c in WIDE])
def test_wide_index_stock_binary_and_parity_target(case):
    """128-bit keys (K = 25).  Inside compareWithDatabase the stock reference compares through a functor declared
    with uint64_t parameters (Compare.hpp:700-706), i.e. on the low words only.  The sequential restatement with
    that quirk switched on reproduces the shipped binary byte for byte -- which pins the 128-bit oracle; with full
    width comparisons it equals the closed form, the parity target of the device, and
    differs from the stock files."""
    stem, kh, kl, frames = case
    d, ix = helpers.load_case("pairs", "idx25")
    assert ix.K == 25
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    texts = {}
    for name, cf, quirk in (("stock", False, True), ("sequential", False, False), ("closed", True, False)):
        res, nq = helpers.oracle_identify(ix, batch, kh, kl, frames, closed_form=cf, cmp64_quirk=quirk)
        texts[name] = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique,
                                     nq, "jsonl", kh, kl, frames, 0.0, 100)
    assert texts["stock"][0] == _read(os.path.join(d, "out_" + stem + ".jsonl"))
    assert texts["stock"][1] == _read(os.path.join(d, "prof_" + stem + ".csv"))
    assert texts["sequential"] == texts["closed"]
    assert texts["closed"][0] != texts["stock"][0]


PAIRED = [("pair", 3), ("pair6", 6)]   # -1 pair_1.fastq -2 pair_2.fastq [--six]


@pytest.mark.parametrize("case", PAIRED, ids=[c[0] for c in PAIRED])
@pytest.mark.parametrize("closed_form", [False, True], ids=["sequential", "closed_form"])
def test_paired_end_byte_identical(case, closed_form):
    """Both mates of a pair enter the batch as two sequences of one read (Read.hpp:834-1049)."""
    stem, frames = case
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_pairs(os.path.join(d, "pair_1.fastq"), os.path.join(d, "pair_2.fastq"))
    assert batch.n == 12 and batch.offsets.shape[0] == 25
    res, nq = helpers.oracle_identify(ix, batch, 12, 7, frames, closed_form=closed_form)
    text, prof = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique,
                                nq, "jsonl", 12, 7, frames, 0.0, 100)
    assert text == _read(os.path.join(d, "out_" + stem + ".jsonl"))
    assert prof == _read(os.path.join(d, "prof_" + stem + ".csv"))


@pytest.mark.parametrize("infile,thr,clean,cont", [("reads.fastq", 0.5, "flt_clean.fastq", "flt_cont.fastq"),
                                                    ("reads.fasta", 0.7, "flta_clean.fasta", "flta_cont.fasta")])
def test_filter_outputs_byte_identical(infile, thr, clean, cont, tmp_path):
    """--filter / --errorThreshold (Compare.hpp:1597-1599, 2448-2596) on the oracle's scores."""
    from kasa_amd import report
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, infile))
    res, _ = helpers.oracle_identify(ix, batch, 12, 7, 3)
    rows = helpers.csr_from_dense(res.M)
    flagged = []
    for r in range(batch.n):
        rk = report.rank_read(rows[r][0], rows[r][1], int(batch.lengths[r]), ix.freq_at(12), 12, 7, 3, 0.0, 100)
        if rk.hits and report.is_contaminant(rk.best, max(h.score for h in rk.hits), thr):
            flagged.append(r)
    report.filter_reads([os.path.join(d, infile)], flagged, str(tmp_path / "c"), str(tmp_path / "x"))
    ext = os.path.splitext(clean)[1]
    assert _read(str(tmp_path / ("c" + ext)), True) == _read(os.path.join(d, clean), True)
    assert _read(str(tmp_path / ("x" + ext)), True) == _read(os.path.join(d, cont), True)


def test_filter_outputs_gzip(tmp_path):
    """--filter --gzip (Compare.hpp:2455,3713-3731): the same two files through zlib, ".gz" appended to their names.  The
    reference's files unpack to the plain ones of the same run; so do the host's."""
    import gzip
    from kasa_amd import report
    d, ix = helpers.load_case("pairs")
    ref_clean, ref_cont = (gzip.open(os.path.join(d, n)).read() for n in ("gflt_clean.fastq.gz", "gflt_cont.fastq.gz"))
    assert ref_clean == _read(os.path.join(d, "flt_clean.fastq"), True) and ref_cont == _read(os.path.join(d, "flt_cont.fastq"), True)
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    res, _ = helpers.oracle_identify(ix, batch, 12, 7, 3)
    rows = helpers.csr_from_dense(res.M)
    flagged = []
    for r in range(batch.n):
        rk = report.rank_read(rows[r][0], rows[r][1], int(batch.lengths[r]), ix.freq_at(12), 12, 7, 3, 0.0, 100)
        if rk.hits and report.is_contaminant(rk.best, max(h.score for h in rk.hits), 0.5):
            flagged.append(r)
    report.filter_reads([os.path.join(d, "reads.fastq")], flagged, str(tmp_path / "c"), str(tmp_path / "x"), gzip_out=True)
    assert gzip.open(str(tmp_path / "c.fastq.gz")).read() == ref_clean
    assert gzip.open(str(tmp_path / "x.fastq.gz")).read() == ref_cont


@pytest.mark.parametrize("closed_form", [False, True], ids=["sequential", "closed_form"])
def test_custom_codon_table(closed_form):
    """-a gc.prt 2 (kASA::setCodonTable, kASA.hpp:579-615) for `build` and `identify`."""
    from oracle import oracle
    d, ix = helpers.load_case("pairs", "idxa")
    lut = oracle.codon_table_from_file(os.path.join(d, "gc.prt"), "2")
    assert (lut != oracle.codon_table()).sum() == 4          # TGA -> W (was ']'), ATA -> M, AGA and AGG -> stop '['
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    res, nq = helpers.oracle_identify(ix, batch, 12, 7, 3, closed_form=closed_form, lut=lut)
    text, prof = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique,
                                nq, "jsonl", 12, 7, 3, 0.0, 100)
    assert text == _read(os.path.join(d, "out_alpha.jsonl"))
    assert prof == _read(os.path.join(d, "prof_alpha.csv"))
