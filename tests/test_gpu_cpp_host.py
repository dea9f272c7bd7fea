"""The C++ host driver (kasa_amd/host/kasa_identify: kASA's `identify` CLI over the C ABI) must write the
same bytes as the reference binary wrote for the same command lines (tests/golden/pairs)."""
import os
import subprocess

import pytest

from kasa_amd import build as hipbuild, capi
from tests.test_oracle_golden import PAIRS, _read, unpack, wants_coverage
from tests import helpers

pytestmark = pytest.mark.gpu

FLAGS = {"json": "--json", "jsonl": "--jsonl", "tsv": "--tsv", "kraken": "--kraken"}


@pytest.mark.parametrize("text", ["device-text", "host-text"])          # kasa_batch_text / the host's Writer over the ranked hits
@pytest.mark.parametrize("case", PAIRS, ids=[c[0] for c in PAIRS])
def test_cpp_host_byte_identical(case, text, tmp_path):
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    stem, infile, fmt, kh, kl, frames, thr, beasts, idx, uniq = unpack(case)
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, idx), "-i", os.path.join(d, infile),
           "-q", out, "-p", prof, FLAGS[fmt], "-b", str(beasts), "-k", str(kh), str(kl), "-m", "4", "-n", "1", "-t", str(tmp_path)]
    if frames == 6:
        cmd.append("--six")
    if frames == 1:
        cmd.append("--one")
    if uniq:
        cmd.append("-e")
    if wants_coverage(case):
        cmd.append("--coverage")
    if thr:
        cmd += ["--threshold", str(thr)]
    if text == "host-text":
        cmd.append("--host-text")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert _read(out) == _read(os.path.join(d, "out_" + stem))
    assert _read(prof) == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))


def test_cpp_host_parameters_file(tmp_path):
    """`--parameters <yaml>` (source/main.cpp:264-302, Utilities.hpp:1114-1400): the reference's config file instead of a command
    line -- the same bytes as the golden command line it stands for (out_six.jsonl: --six --jsonl -b 3)."""
    exe = hipbuild.build_host()
    case = next(c for c in PAIRS if unpack(c)[5] == 6 and unpack(c)[2] == "jsonl" and not unpack(c)[9] and unpack(c)[8] == "idx")
    stem, infile, fmt, kh, kl, frames, thr, beasts, idx, uniq = unpack(case)
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    y = tmp_path / "config.yaml"
    y.write_text(f"""# kASA config
Mode: identify
Index: "{os.path.join(d, idx)}"
ContentFile: {os.path.join(d, 'content.txt')}
kHigh: {kh}
kLow: {kl}
NumberOfThreads: 1
AvailableRAMinGB: 4
FilePathForTemporaryFiles: {tmp_path}
InputFileOrFolder: {os.path.join(d, infile)}
Six: true
Three: false
ProfileOutputfile: {prof}
ReadIDtoTaxIDOutputfile: {out}
ReadIDtoTaxIDOutputFormat: jsonl
NumberOfTaxaPerRead: {beasts}
UniqueKmersOnly: false
ThresholdForScore: {thr if thr else 0}
Filter: _ _
ShrinkingStrategy: 2
""")
    r = subprocess.run([exe, "--parameters", str(y)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert _read(out) == _read(os.path.join(d, "out_" + stem))
    assert _read(prof) == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))
    r = subprocess.run([exe, "--parameters", str(tmp_path / "nope.yaml")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 1 and "Config file not found" in r.stderr


def test_cpp_host_errors_like_the_reference(tmp_path):
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    r = subprocess.run([exe, "identify", "-d", os.path.join(d, "nope"), "-i", os.path.join(d, "reads.fastq")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 1 and r.stderr.startswith("ERROR: ")
    r = subprocess.run([exe, "identify", "--frobnicate"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 1 and "unknown parameter" in r.stderr


def test_cpp_host_wide_index(tmp_path):
    """128-bit index through the C++ driver: default k range (25..7, main.cpp:1065-1067) and -k 12 7, against the
    oracle's closed form rendered by the Python host writers."""
    import os as _os
    from kasa_amd import reads
    from tests.test_oracle_golden import WIDE
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    d, ix = helpers.load_case("pairs", "idx25")
    batch = reads.parse_reads(_os.path.join(d, "reads.fastq"))
    for stem, kh, kl, frames in WIDE:
        out, prof = str(tmp_path / ("out_" + stem)), str(tmp_path / ("prof_" + stem))
        cmd = [exe, "identify", "-c", _os.path.join(d, "content.txt"), "-d", _os.path.join(d, "idx25"), "-i",
               _os.path.join(d, "reads.fastq"), "-q", out, "-p", prof, "--jsonl", "-b", "100", "-n", "1"]
        if (kh, kl) != (25, 7):
            cmd += ["-k", str(kh), str(kl)]
        if frames == 6:
            cmd.append("--six")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        res, nq = helpers.oracle_identify(ix, batch, kh, kl, frames, closed_form=True)
        text, ptext = helpers.render(ix, batch, helpers.csr_from_dense(res.M), res.count_all, res.count_unique, nq,
                                     "jsonl", kh, kl, frames, 0.0, 100)
        assert _read(out) == text
        assert _read(prof) == ptext
        # ... and the same over a partitioned index (64-byte records through the partitions' contexts)
        n_rec = int(open(_os.path.join(d, "idx25_info.txt")).read().split()[0])
        r = subprocess.run(cmd + ["-v"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                           env=dict(_os.environ, KASA_INDEX_PART_RECORDS=str(n_rec // 4 + 1)))
        assert r.returncode == 0, r.stderr
        assert "partitions on every device" in r.stdout
        assert _read(out) == text
        assert _read(prof) == ptext


def test_cpp_host_paired_end(tmp_path):
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    for stem, extra in (("pair", []), ("pair6", ["--six"])):
        out, prof = str(tmp_path / ("o_" + stem)), str(tmp_path / ("p_" + stem))
        cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-1", os.path.join(d, "pair_1.fastq"),
               "-2", os.path.join(d, "pair_2.fastq"), "-q", out, "-p", prof, "--jsonl", "-b", "100"] + extra
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert _read(out) == _read(os.path.join(d, "out_" + stem + ".jsonl"))
        assert _read(prof) == _read(os.path.join(d, "prof_" + stem + ".csv"))


@pytest.mark.parametrize("infile,stem", [("reads.fastq", "b100.jsonl"), ("reads.fasta", "fasta.jsonl"),
                                          ("edge_crlf.fasta", "edge_crlf.jsonl"), ("edge_multi.fastq", "edge_multi.jsonl")])
def test_cpp_host_parallel_parser_and_writer(infile, stem, tmp_path):
    """The input cut into many small runs of records parsed by 4 threads, the text written by 4 threads: same bytes."""
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, infile),
           "-q", out, "-p", prof, "--jsonl", "-b", "100", "-n", "4"]
    # (and the device's text fetched through page-locked buffers of 1000 bytes that take turns)
    env = dict(os.environ, KASA_PARSE_CHUNK="700", KASA_TEXT_PIECE="1000")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert _read(out) == _read(os.path.join(d, "out_" + stem))
    r = subprocess.run(cmd + ["--host-text"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert _read(out) == _read(os.path.join(d, "out_" + stem))
    assert _read(prof) == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))


@pytest.mark.parametrize("infile,extra,clean,cont", [("reads.fastq", [], "flt_clean.fastq", "flt_cont.fastq"),
                                                      ("reads.fasta", ["--errorThreshold", "0.7"], "flta_clean.fasta", "flta_cont.fasta")])
def test_cpp_host_filter(infile, extra, clean, cont, tmp_path):
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    ext = os.path.splitext(clean)[1]
    for with_q in (True, False):                                   # --filter works without -q as well (Compare.hpp:2878)
        c, x = str(tmp_path / ("c%d" % with_q)), str(tmp_path / ("x%d" % with_q))
        cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, infile),
               "-p", str(tmp_path / "prof.csv"), "--jsonl", "-b", "100", "--filter", c, x] + extra
        if with_q:
            cmd += ["-q", str(tmp_path / "out.jsonl")]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert _read(c + ext, True) == _read(os.path.join(d, clean), True)
        assert _read(x + ext, True) == _read(os.path.join(d, cont), True)


def test_cpp_host_custom_codon_table(tmp_path):
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "o"), str(tmp_path / "p")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idxa"), "-i", os.path.join(d, "reads.fastq"),
           "-a", os.path.join(d, "gc.prt"), "2", "-q", out, "-p", prof, "--jsonl", "-b", "100"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert _read(out) == _read(os.path.join(d, "out_alpha.jsonl"))
    assert _read(prof) == _read(os.path.join(d, "prof_alpha.csv"))


def test_cpp_host_several_batches(tmp_path):
    """The input cut into many device batches: the profile is the same file (it is additive over batches), the per-read
    file names the same taxa in the same order (scores may move in the last float digit with the batch, as they do in
    the reference when -m changes)."""
    import re
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "o"), str(tmp_path / "p")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq"),
           "-q", out, "-p", prof, "--jsonl", "-b", "100"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, KASA_MAX_BATCH_KMERS="1500", KASA_TEXT_PIECE="3000"))
    assert r.returncode == 0, r.stderr
    assert _read(prof) == _read(os.path.join(d, "prof_b100.csv"))
    strip = lambda t: re.sub(r'"(k-mer Score|Relative Score|Error)": [-0-9.e+infa]+', r'"\1": x', t)
    assert strip(_read(out)) == strip(_read(os.path.join(d, "out_b100.jsonl")))


def _run_host(args, env=None, timeout=600):
    exe = hipbuild.build_host()
    r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    return r


@pytest.mark.parametrize("infile,stem", [("reads.fastq", "b100"), ("reads.fasta", "fasta"), ("edge_noeol.fasta", "edge_noeol")])
def test_cpp_host_streams_the_input_in_chunks(infile, stem, tmp_path):
    """N1: the input is read block by block, cut at record boundaries and parsed by several threads; tiny blocks here so
    that the golden input spans dozens of chunks.  Same bytes as the reference."""
    assert capi.device_count() > 0
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    _run_host(["identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, infile), "-q", out, "-p", prof,
               "--jsonl", "-b", "100", "-n", "4"], env={"KASA_READ_BLOCK": "1500", "KASA_PARSE_CHUNK": "400"})
    assert _read(out) == _read(os.path.join(d, "out_" + stem + ".jsonl"))
    assert _read(prof) == _read(os.path.join(d, "prof_" + stem + ".csv"))


@pytest.mark.parametrize("block", ["1500", None])
def test_cpp_host_gzipped_input(block, tmp_path):
    """The same reads gzip'ed (two members, as `cat a.gz b.gz` leaves them): inflated as they come, block by block."""
    import gzip
    assert capi.device_count() > 0
    d = os.path.join(helpers.GOLDEN, "pairs")
    raw = open(os.path.join(d, "reads.fastq"), "rb").read()
    cut = raw.index(b"\n@", len(raw) // 2) + 1
    gz = str(tmp_path / "reads.fastq.gz")
    with open(gz, "wb") as f:
        f.write(gzip.compress(raw[:cut]) + gzip.compress(raw[cut:]))
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    env = {"KASA_READ_BLOCK": block, "KASA_PARSE_CHUNK": "400"} if block else {}
    _run_host(["identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", gz, "-q", out, "-p", prof,
               "--jsonl", "-b", "100", "-n", "3"], env=env)
    assert _read(out) == _read(os.path.join(d, "out_b100.jsonl"))
    assert _read(prof) == _read(os.path.join(d, "prof_b100.csv"))


def test_cpp_host_identify_multiple(tmp_path):
    """identify_multiple (main.cpp:1118-1334): several input files as a job queue over ONE shared index object, one
    context per worker; outputs named <prefix><file name><format ending> / <prefix><file name>.csv."""
    import shutil
    assert capi.device_count() > 0
    d = os.path.join(helpers.GOLDEN, "pairs")
    ind = tmp_path / "in"
    ind.mkdir()
    src = {"sampleA.fastq": ("reads.fastq", "b100"), "sampleB.fasta": ("reads.fasta", "fasta"), "sampleC.fastq": ("reads_dup.fastq", "dup"),
           "sampleD.fasta": ("exampleInput.fasta", "exampleInput")}
    cases = {}
    for f, (orig, stem) in src.items():
        shutil.copy(os.path.join(d, orig), str(ind / f))
        cases[f] = stem
    _run_host(["identify_multiple", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", str(ind) + "/",
               "-q", str(tmp_path / "rtt_"), "-p", str(tmp_path / "prof_"), "--jsonl", "-b", "100", "-n", "2"])
    for f, stem in cases.items():
        name = f.rsplit(".", 1)[0]
        assert _read(str(tmp_path / ("rtt_" + name + ".jsonl"))) == _read(os.path.join(d, "out_" + stem + ".jsonl")), f
        assert _read(str(tmp_path / ("prof_" + name + ".csv"))) == _read(os.path.join(d, "prof_" + stem + ".csv")), f


def test_cpp_host_rccl_reduce_with_one_rank(tmp_path):
    """kasa_profile_allreduce (limbs packed on the device, ncclAllReduce, carries folded back) with a one-rank RCCL
    communicator: the profile must be what it was.  More ranks need more GPUs than this box has."""
    assert capi.device_count() > 0
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    _run_host(["identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq"), "-q", out, "-p", prof,
               "--jsonl", "-b", "100", "-n", "2", "--devices", "0", "--coverage"], env={"KASA_FORCE_ALLREDUCE": "1"})
    assert _read(out) == _read(os.path.join(d, "out_cov.jsonl"))
    assert _read(prof) == _read(os.path.join(d, "prof_cov.csv"))


def test_cpp_host_filter_gzip(tmp_path):
    """--filter --gzip: the driver's files unpack to what the reference's unpack to."""
    import gzip
    assert capi.device_count() > 0
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    c, x = str(tmp_path / "c"), str(tmp_path / "x")
    r = subprocess.run([exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq"),
                        "-p", str(tmp_path / "prof.csv"), "--jsonl", "-b", "100", "--filter", c, x, "--gzip"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for mine, ref in ((c + ".fastq.gz", "gflt_clean.fastq.gz"), (x + ".fastq.gz", "gflt_cont.fastq.gz")):
        assert open(mine, "rb").read(2) == b"\x1f\x8b"
        assert gzip.open(mine).read() == gzip.open(os.path.join(d, ref)).read()


@pytest.mark.parametrize("fmt,extra", [("--jsonl", []), ("--tsv", ["--six"]), ("--json", ["-b", "7", "--threshold", "0.02"])])
def test_cpp_host_paths_agree_at_scale(fmt, extra, tmp_path):
    """150 000 reads of 1..400 bases (reads too short for a k-mer, runs of N, lower case), several device batches: the text
    written on the device, the host's writer over the device's ranking (--host-text) and the host's own ranking over the
    full rows (--host-rank) must give the same file and the same profile."""
    import numpy as np
    from kasa_amd import formats, synth
    assert capi.device_count() > 0
    g = synth.genomes(40, 20_000, seed=3)
    ix = synth.index_from_genomes(g)
    formats.write_index(ix, str(tmp_path / "idx"), str(tmp_path / "content.txt"))
    rng = np.random.default_rng(17)
    n = 150_000
    flat = g.reshape(-1)
    lens = rng.integers(1, 401, size=n)
    lens[rng.random(n) < 0.6] = 150
    starts = rng.integers(0, flat.shape[0] - 400, size=n)
    lines = []
    for r in range(n):
        s = flat[starts[r]:starts[r] + lens[r]].copy()
        if r % 11 == 0 and lens[r] > 20:
            a = int(rng.integers(0, lens[r] - 5)); s[a:a + int(rng.integers(1, 6))] = ord("N")
        if r % 13 == 0:
            s = np.frombuffer(bytes(s).lower(), dtype=np.uint8)
        lines.append(b"@r%d some text\n" % r + bytes(s) + b"\n+\n" + b"I" * int(lens[r]) + b"\n")
    fq = str(tmp_path / "reads.fastq")
    open(fq, "wb").write(b"".join(lines))
    outs = []
    for mode in ([], ["--host-text"], ["--host-rank"]):
        out, prof = str(tmp_path / ("out" + "".join(mode))), str(tmp_path / ("prof" + "".join(mode)))
        _run_host(["identify", "-c", str(tmp_path / "content.txt"), "-d", str(tmp_path / "idx"), "-i", fq, "-q", out, "-p", prof, fmt, "-m", "4"] + extra + mode,
                  env={"KASA_MAX_BATCH_KMERS": "6000000", "KASA_TEXT_PIECE": "3000000"})
        outs.append((open(out, "rb").read(), open(prof, "rb").read()))
    assert outs[0][1] == outs[1][1] == outs[2][1]
    assert outs[0][0] == outs[1][0], "device text != host text over the device's ranking"
    assert outs[0][0] == outs[2][0], "device ranking + text != host ranking + text"
    assert outs[0][0].count(b"\n") > n // 2


@pytest.mark.parametrize("case", PAIRS, ids=[c[0] for c in PAIRS])
def test_cpp_host_partitioned_index_byte_identical(case, tmp_path):
    """An index of 2^32 records and more runs as range partitions on the device (Compare.hpp:286-318 streams any size); with
    KASA_INDEX_PART_RECORDS the driver cuts the small golden indexes the same way -- several partitions, the same bytes."""
    exe = hipbuild.build_host()
    stem, infile, fmt, kh, kl, frames, thr, beasts, idx, uniq = unpack(case)
    d = os.path.join(helpers.GOLDEN, "pairs")
    out, prof = str(tmp_path / "out"), str(tmp_path / "prof.csv")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, idx), "-i", os.path.join(d, infile),
           "-q", out, "-p", prof, FLAGS[fmt], "-b", str(beasts), "-k", str(kh), str(kl), "-m", "4", "-n", "1", "-t", str(tmp_path), "-v"]
    cmd += (["--six"] if frames == 6 else []) + (["--one"] if frames == 1 else []) + (["-e"] if uniq else [])
    cmd += (["--coverage"] if wants_coverage(case) else []) + (["--threshold", str(thr)] if thr else [])
    n_rec = int(open(os.path.join(d, idx + "_info.txt")).read().split()[0])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, KASA_INDEX_PART_RECORDS=str(n_rec // 5 + 1)))
    assert r.returncode == 0, r.stderr
    parts = [int(l.split()[6]) for l in r.stdout.splitlines() if l.startswith("OUT: Index of")]
    assert parts and parts[0] >= 5, r.stdout
    assert _read(out) == _read(os.path.join(d, "out_" + stem))
    assert _read(prof) == _read(os.path.join(d, "prof_" + stem.rsplit(".", 1)[0] + ".csv"))


def test_cpp_host_partitioned_index_across_batches_and_pieces(tmp_path):
    """... and with the reference's batches and a sequence read in pieces on top (tests/golden/batches: m1 and long)."""
    import gzip
    import lzma
    import shutil
    exe = hipbuild.build_host()
    src = os.path.join(helpers.GOLDEN, "batches")
    d = str(tmp_path)
    for f in ("content.txt.gz", "idx_f.txt.gz"):
        with gzip.open(os.path.join(src, f), "rb") as g, open(os.path.join(d, f[:-3]), "wb") as o:
            shutil.copyfileobj(g, o)
    for f in ("idx", "idx_info.txt", "idx_trie", "idx_trie.txt", "reads.fastq.gz"):
        shutil.copy(os.path.join(src, f), os.path.join(d, f))
    with lzma.open(os.path.join(src, "long.fasta.xz"), "rb") as g, open(os.path.join(d, "long.fasta"), "wb") as o:
        shutil.copyfileobj(g, o)
    for infile, gold in (("reads.fastq.gz", "m1"), ("long.fasta", "long")):
        out, prof = os.path.join(d, "out.jsonl"), os.path.join(d, "prof.csv")
        cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, infile), "-q", out, "-p", prof,
               "--jsonl", "-b", "100", "-m", "1", "-n", "1", "-v"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=dict(os.environ, KASA_INDEX_PART_RECORDS="30000"))
        assert r.returncode == 0, r.stderr
        assert "partitions on every device" in r.stdout
        with gzip.open(os.path.join(src, "out_%s.jsonl.gz" % gold), "rb") as f:
            assert _read(out) == f.read().decode("latin-1")
        assert _read(prof) == _read(os.path.join(src, "prof_%s.csv" % gold))
