#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the development container, where /root/reference exists.  The reference's own test
assets hold no expected outputs for `identify`, and its source tree cannot be
compiled here under the round's rules (it needs cmake-generated STXXL headers) -- but it ships a
prebuilt binary of the same version, binaries/kASA_linux (v1.4.9, an AVX build), which runs here.
This script drives that binary: `build` on a seeded synthetic database, then `identify` in several
configurations, and stores inputs + the binary's outputs as fixtures.  Nothing of the reference's
source or binary is copied; only its outputs (and the index files it wrote, which are data).

The binary is started through the dynamic loader because the read-only mount drops the x bit.

    python tests/golden/make_fixtures.py          # regenerates every case
"""
import gzip
import json
import lzma
import os
import random
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference"
KASA = ["/lib64/ld-linux-x86-64.so.2", os.path.join(REF, "binaries", "kASA_linux")]
HERE = os.path.dirname(os.path.abspath(__file__))


def run(args, cwd):
    tmp = os.path.join(cwd, "tmp")
    os.makedirs(tmp, exist_ok=True)
    cmd = KASA + args + ["-t", tmp + "/"]
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    if p.returncode != 0:
        sys.stderr.write(p.stdout)
        raise SystemExit("reference failed: " + " ".join(args))
    return p.stdout


def mutate(s, rate, rng):
    s = list(s)
    for i in range(len(s)):
        if rng.random() < rate:
            s[i] = rng.choice("ACGT")
    return "".join(s)


def write_db(d, genomes):
    with open(os.path.join(d, "db.fasta"), "w") as f, open(os.path.join(d, "content.txt"), "w") as c:
        for g, s in enumerate(genomes):
            acc = "ACC%03d.1" % g
            f.write(">%s synthetic taxon %d\n" % (acc, g))
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
            c.write("Taxon %d\t%d\t%d\t%s\n" % (g, 100 + g, 100 + g, acc))


def trim_index(d):
    """The reference zero-pads index and trie files to 2,101,248-byte blocks; keep the records only."""
    jobs = []
    for stem, rec in (("idx", 12), ("idx_half", 6), ("idx25", 20), ("idxa", 12)):
        if not os.path.exists(os.path.join(d, stem + "_info.txt")):
            continue
        n = int(open(os.path.join(d, stem + "_info.txt")).read().split()[0])
        m = int(open(os.path.join(d, stem + "_trie.txt")).read().split()[0])
        jobs += [(stem, n * rec), (stem + "_trie", m * 12)]
    for name, nbytes in jobs:
        p = os.path.join(d, name)
        with open(p, "rb") as f:
            data = f.read(nbytes)
        with open(p, "wb") as f:
            f.write(data)


# standard genetic code in kASA's letters: stops TAA/TAG -> '*' (written '[' by the reader), TGA -> ']'
# (the table `build` translates the database with; README "amino-acid-like" alphabet)
_AA = {}
for _i, _c0 in enumerate("TCAG"):
    for _j, _c1 in enumerate("TCAG"):
        for _k, _c2 in enumerate("TCAG"):
            _AA[_c0 + _c1 + _c2] = "FFLLSSSSYY**CC]WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"[16 * _i + 4 * _j + _k]


def translate(s):
    return "".join(_AA[s[i:i + 3]] for i in range(0, len(s) - 2, 3))


def write_protein_reads(out, genomes):
    rng = random.Random(13)
    L = len(genomes[0])
    reads = []
    for r in range(24):
        g = rng.randrange(len(genomes)); p = rng.randrange(L - 180); fr = rng.randrange(3)
        ln = rng.choice([30, 50, 60])
        reads.append(("prot%d_t%d" % (r, g), translate(genomes[g][p + fr:p + fr + 3 * ln])))
    g0 = genomes[0]
    reads.append(("plower", translate(g0[300:450]).lower()))
    reads.append(("ptiny5", translate(g0[600:615])))
    reads.append(("plen12", translate(g0[600:636])))
    reads.append(("plen13", translate(g0[600:639])))
    reads.append(("plen14", translate(g0[600:642])))
    reads.append(("pforeign", "".join(rng.choice("ACDEFGHIKLMNPQRSTVWY") for _ in range(50))))
    with open(os.path.join(out, "reads_prot.fasta"), "w") as f:
        for n, s in reads:
            f.write(">%s\n%s\n" % (n, s))


def case_pairs(out):
    """6 taxa in 3 sibling pairs (3 % apart): at most 2-3 taxa per k-mer -> the binary's scalar path."""
    rng = random.Random(11)
    G, L = 6, 1500
    genomes = []
    for g in range(G):
        genomes.append(mutate(genomes[g - 1], 0.03, rng) if g % 2 else
                       "".join(rng.choice("ACGT") for _ in range(L)))
    write_db(out, genomes)
    rng = random.Random(12)
    reads = []
    for r in range(48):
        g = rng.randrange(G)
        p = rng.randrange(L - 150)
        reads.append(("read%d_t%d" % (r, g), mutate(genomes[g][p:p + 150], 0.01, rng)))
    g0 = genomes[0]
    reads.append(("withN", g0[100:160] + "NNNN" + g0[164:250]))
    reads.append(("dash_and_lower", g0[300:340].lower() + "-" + g0[341:450]))
    reads.append(("tiny10", g0[500:510]))                       # padded to 3K -> zero k-mers
    reads.append(("len21", g0[520:541]))                        # padded length 36 -> 0 k-mers
    reads.append(("len22", g0[520:542]))
    reads.append(("len23", g0[520:543]))                        # first length with k-mers
    reads.append(("len36", g0[600:636]))
    reads.append(("len40", g0[600:640]))
    reads.append(("tandem", g0[700:730] * 5))                   # the same k-mers several times in one read
    reads.append(("twice", g0[800:875] + g0[800:875]))
    reads.append(("foreign", "".join(rng.choice("ACGT") for _ in range(150))))
    reads.append(("polyA", "A" * 150))
    with open(os.path.join(out, "reads.fastq"), "w") as f:
        for n, s in reads:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    with open(os.path.join(out, "reads.fasta"), "w") as f:      # multi-line FASTA of the same reads
        for n, s in reads:
            f.write(">%s\n" % n)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    g = genomes
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    dup = [("tandem", g[0][700:730] * 5), ("twice", g[0][800:875] * 2), ("triple", g[2][100:150] * 3),
           ("plain1", g[1][0:150]), ("plain3", g[3][200:350]), ("plain4", g[4][400:550]), ("plain5", g[5][900:1050]),
           ("palin", g[2][300:360] + "".join(comp[c] for c in reversed(g[2][300:360])))]
    with open(os.path.join(out, "reads_dup.fastq"), "w") as f:
        for n, s in dup:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    write_protein_reads(out, genomes)
    # input formats at the edges of the reader: CRLF line ends (the reference keeps the '\r': it lands in the name, in
    # "Length" and, as a non-ACGT letter, in the k-mers), a header with spaces and '+name' / '#' quality lines, a last
    # line without line feed
    with open(os.path.join(out, "edge_crlf.fasta"), "w", newline="") as f:
        f.write(">r1 desc\r\n" + g[0][0:80] + "\r\n" + g[0][80:150] + "\r\n>r2\r\n" + g[1][200:350] + "\r\n")
    with open(os.path.join(out, "edge_multi.fastq"), "w") as f:
        f.write("@q1\n" + g[3][0:150] + "\n+\n" + "I" * 150 + "\n@q2 x y\n" + g[4][10:100] + "\n+q2\n" + "#" * 90 + "\n")
    with open(os.path.join(out, "edge_noeol.fasta"), "w") as f:
        f.write(">r1\n" + g[5][0:150] + "\n>r2\n" + g[5][300:450])
    # paired-end input: mates 100 bp each, 150 bp apart, second mate reverse-complemented
    prng = random.Random(31)
    with open(os.path.join(out, "pair_1.fastq"), "w") as f1, open(os.path.join(out, "pair_2.fastq"), "w") as f2:
        for r in range(12):
            gi = prng.randrange(G)
            p = prng.randrange(L - 400)
            a = genomes[gi][p:p + 100]
            b = "".join(comp[c] for c in reversed(genomes[gi][p + 250:p + 350]))
            f1.write("@pair%d/1\n%s\n+\n%s\n" % (r, a, "I" * 100))
            f2.write("@pair%d/2\n%s\n+\n%s\n" % (r, b, "I" * 100))
    run(["build", "-c", "content.txt", "-d", "idx", "-i", "db.fasta", "-m", "4", "-n", "1"], out)
    base = ["identify", "-c", "content.txt", "-d", "idx", "-m", "4", "-n", "1"]
    runs = {
        "default.json": ["-i", "reads.fastq", "--json"],
        "b100.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100"],
        "b100.tsv": ["-i", "reads.fastq", "--tsv", "-b", "100"],
        "b1.tsv": ["-i", "reads.fastq", "--tsv", "-b", "1"],
        "default.ktsv": ["-i", "reads.fastq", "--kraken"],
        "fasta.jsonl": ["-i", "reads.fasta", "--jsonl", "-b", "100"],
        "k12_9.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "-k", "12", "9"],
        "k10_7.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "-k", "10", "7"],
        "k12_12.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "-k", "12", "12"],
        "six.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "--six"],
        "thr04.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "--threshold", "0.4"],
        "ram.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "-r"],
        "one.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "--one"],
        # reads that repeat their own k-mers, without and with -e (std::unique after an unstable sort: the outcome
        # is defined only where no OTHER read shares the repeated k-mers, hence the disjoint windows)
        "dup.jsonl": ["-i", "reads_dup.fastq", "--jsonl", "-b", "100"],
        "unique.jsonl": ["-i", "reads_dup.fastq", "--jsonl", "-b", "100", "-e"],
        "unique6.jsonl": ["-i", "reads_dup.fastq", "--jsonl", "-b", "100", "-e", "--six"],
        # amino-acid input (the binary detects it from the first sequence)
        "prot.jsonl": ["-i", "reads_prot.fasta", "--jsonl", "-b", "100"],
        "edge_crlf.jsonl": ["-i", "edge_crlf.fasta", "--jsonl", "-b", "100"],
        "edge_multi.jsonl": ["-i", "edge_multi.fastq", "--jsonl", "-b", "100"],
        "edge_noeol.jsonl": ["-i", "edge_noeol.fasta", "--jsonl", "-b", "100"],
        "cov.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "--coverage"],
        "pair.jsonl": ["-1", "pair_1.fastq", "-2", "pair_2.fastq", "--jsonl", "-b", "100"],
        "pair6.jsonl": ["-1", "pair_1.fastq", "-2", "pair_2.fastq", "--jsonl", "-b", "100", "--six"],
    }
    for name, extra in runs.items():
        stem = name.rsplit(".", 1)[0]
        run(base + extra + ["-q", "out_" + name, "-p", "prof_" + stem + ".csv"], out)
    # --filter: the input split into clean reads and contaminants (default --errorThreshold 0.5, and 0.7 on the FASTA)
    run(base + ["-i", "reads.fastq", "--jsonl", "-b", "100", "--filter", "flt_clean", "flt_cont", "-q", "out_flt.jsonl", "-p", "prof_flt.csv"], out)
    run(base + ["-i", "reads.fasta", "--jsonl", "-b", "100", "--filter", "flta_clean", "flta_cont", "--errorThreshold", "0.7",
                "-q", "out_flta.jsonl", "-p", "prof_flta.csv"], out)
    for junk in ("out_flt.jsonl", "prof_flt.csv", "out_flta.jsonl", "prof_flta.csv"):   # same as b100 / fasta
        os.remove(os.path.join(out, junk))
    # --gzip: the filter's files written through the reference's zlib (same split as flt_*)
    run(base + ["-i", "reads.fastq", "--jsonl", "-b", "100", "--filter", "gflt_clean", "gflt_cont", "--gzip", "-q", "out_gflt.jsonl", "-p", "prof_gflt.csv"], out)
    for junk in ("out_gflt.jsonl", "prof_gflt.csv"):
        os.remove(os.path.join(out, junk))
    # --coherence (Compare::postProcess): one more column / field per read in every text format but the Kraken one; with
    # --six on an input whose last read matches on both strands (reads_dup.fastq ends in a palindrome) ...
    coh = {
        "coh.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100"],
        "coh.tsv": ["-i", "reads.fastq", "--tsv", "-b", "3"],
        "coh.json": ["-i", "reads.fastq", "--json"],
        "coh.ktsv": ["-i", "reads.fastq", "--kraken"],
        "coh_fasta.tsv": ["-i", "reads.fasta", "--tsv", "-b", "100"],
        "coh_one.jsonl": ["-i", "reads.fastq", "--jsonl", "-b", "100", "--one"],
        "coh_k12_9.tsv": ["-i", "reads.fastq", "--tsv", "-b", "100", "-k", "12", "9"],
        "coh_prot.jsonl": ["-i", "reads_prot.fasta", "--jsonl", "-b", "100"],
        "coh_dup.tsv": ["-i", "reads_dup.fastq", "--tsv", "-b", "100"],
        "coh_dup6.tsv": ["-i", "reads_dup.fastq", "--tsv", "-b", "100", "--six"],
    }
    for name, extra in coh.items():
        run(base + extra + ["--coherence", "-q", "out_" + name, "-p", "prof_" + name.rsplit(".", 1)[0] + ".csv"], out)
    # ... and on one where the reference's walk runs off the end of its vector after the last strand switch: it ends with
    # an exception (vector::at); what it printed is the fixture
    p = subprocess.run(KASA + base + ["-i", "reads.fastq", "--tsv", "--six", "--coherence", "-q", "out_coh_six_throws.tsv", "-p",
                                      "prof_coh_six_throws.csv", "-t", os.path.join(out, "tmp") + "/"], cwd=out, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=120)
    with open(os.path.join(out, "coh_six_throws.err"), "w") as f:
        f.write("".join(l + "\n" for l in p.stdout.splitlines() if l.startswith("ERROR: vector")))
    for junk in ("out_coh_six_throws.tsv", "prof_coh_six_throws.csv"):
        if os.path.exists(os.path.join(out, junk)):
            os.remove(os.path.join(out, junk))
    # --filter with --coherence: a read also goes to the contaminants when its coherence reaches --coherenceThreshold
    run(base + ["-i", "reads.fastq", "--jsonl", "-b", "100", "--coherence", "--coherenceThreshold", "11.99", "--errorThreshold", "0.46",
                "--filter", "cflt_clean", "cflt_cont", "-q", "out_cflt.jsonl", "-p", "prof_cflt.csv"], out)
    for junk in ("out_cflt.jsonl", "prof_cflt.csv"):
        os.remove(os.path.join(out, junk))
    # the "halved" index of shrink strategy 2 (6-byte records) and a run on it
    run(["shrink", "-c", "content.txt", "-d", "idx", "-o", "idx_half", "-s", "2", "-m", "4", "-n", "1"], out)
    run(base + ["-d", "idx_half", "-i", "reads.fastq", "--jsonl", "-b", "100", "-q", "out_half.jsonl", "-p", "prof_half.csv"], out)
    # a 128-bit index (k up to 25) of the same database and runs on it
    run(["build", "-c", "content.txt", "-d", "idx25", "-i", "db.fasta", "-m", "4", "-n", "1", "--kH", "25"], out)
    wide = {"w25_7": ["-k", "25", "7"], "w12_7": ["-k", "12", "7"], "w25_20": ["-k", "25", "20"], "w16_9": ["-k", "16", "9"],
            "w25_7_six": ["-k", "25", "7", "--six"]}
    for name, extra in wide.items():
        run(["identify", "-c", "content.txt", "-d", "idx25", "-m", "4", "-n", "1", "--jsonl", "-b", "100", "-i", "reads.fastq"]
            + extra + ["-q", "out_" + name + ".jsonl", "-p", "prof_" + name + ".csv"], out)
    run(["identify", "-c", "content.txt", "-d", "idx25", "-m", "4", "-n", "1", "--tsv", "-b", "100", "-i", "reads.fastq", "-k", "25", "7",
         "--coherence", "-q", "out_coh_w25_7.tsv", "-p", "prof_coh_w25_7.csv"], out)
    # a custom translation table (-a <gc.prt> <id>): index built and queried with the vertebrate mitochondrial code
    with open(os.path.join(out, "gc.prt"), "w") as f:
        f.write(GC_PRT)
    run(["build", "-c", "content.txt", "-d", "idxa", "-i", "db.fasta", "-m", "4", "-n", "1", "-a", "gc.prt", "2"], out)
    run(["identify", "-c", "content.txt", "-d", "idxa", "-m", "4", "-n", "1", "--jsonl", "-b", "100", "-i", "reads.fastq",
         "-a", "gc.prt", "2", "-q", "out_alpha.jsonl", "-p", "prof_alpha.csv"], out)
    # the reference's own example input (2 reads; N- and '-'-containing, multi-line FASTA)
    shutil.copy(os.path.join(REF, "example/work/input/exampleInput.fasta"), os.path.join(out, "exampleInput.fasta"))
    run(base + ["-i", "exampleInput.fasta", "--jsonl", "-b", "100", "-q", "out_exampleInput.jsonl",
                "-p", "prof_exampleInput.csv"], out)
    # (example.fastq.gz is left out: the shipped binary spins forever on that gzipped input here)


GC_PRT = """--  Genetic code tables in the layout of NCBI's gc.prt (written for the tests; three of the standard tables)
Genetic-code-table ::= {
 {
  name "Standard" ,
  name "SGC0" ,
  id 1 ,
  ncbieaa  "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
  sncbieaa "---M------**--*----M---------------M----------------------------"
  -- Base1  TTTTTTTTTTTTTTTTCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAAGGGGGGGGGGGGGGGG
  -- Base2  TTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGG
  -- Base3  TCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAG
 },
 {
  name "Vertebrate Mitochondrial" ,
  name "SGC1" ,
  id 2 ,
  ncbieaa  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG",
  sncbieaa "----------**--------------------MMMM----------**---M------------"
  -- Base1  TTTTTTTTTTTTTTTTCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAAGGGGGGGGGGGGGGGG
  -- Base2  TTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGG
  -- Base3  TCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAG
 },
 {
  name "Mold Mitochondrial; Protozoan Mitochondrial" ,
  name "SGC3" ,
  id 4 ,
  ncbieaa  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",
  sncbieaa "--MM------**-------M------------MMMM---------------M------------"
  -- Base1  TTTTTTTTTTTTTTTTCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAAGGGGGGGGGGGGGGGG
  -- Base2  TTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGG
  -- Base3  TCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAG
 }
}
"""


def case_clones(out):
    """6 near-identical taxa: most k-mers are shared by > 3 taxa, so the shipped (AVX) binary takes
    its scoreMatchAVX branch.  Reads sit on disjoint windows so no two reads share a k-mer."""
    rng = random.Random(21)
    L = 1200
    base_g = "".join(rng.choice("ACGT") for _ in range(L))
    genomes = [base_g] + [mutate(base_g, 0.004, rng) for _ in range(5)]
    write_db(out, genomes)
    rng = random.Random(22)
    with open(os.path.join(out, "reads.fastq"), "w") as f:
        for r in range(7):
            g = rng.randrange(6)
            p = r * 160
            s = genomes[g][p:p + 150]
            f.write("@clone%d_t%d\n%s\n+\n%s\n" % (r, g, s, "I" * 150))
        # the same k-mers several times inside ONE read: groups with > 3 taxa and > 1 hit, where the
        # AVX branch loses increments (and tie order cannot matter: all hits carry the same read id)
        for r, unit in enumerate((30, 45)):
            s = (genomes[0][1125:1125 + unit] * 6)[:150]
            f.write("@tandem%d\n%s\n+\n%s\n" % (r, s, "I" * 150))
    run(["build", "-c", "content.txt", "-d", "idx", "-i", "db.fasta", "-m", "4", "-n", "1"], out)
    base = ["identify", "-c", "content.txt", "-d", "idx", "-m", "4", "-n", "1", "-i", "reads.fastq"]
    run(base + ["--jsonl", "-b", "100", "-q", "out_b100.jsonl", "-p", "prof_b100.csv"], out)
    run(base + ["-p", "prof_only.csv"], out)


def build_probe():
    """tests/golden/batch_probe.c -> a preload library that counts the reads of every reference batch."""
    so = os.path.join(tempfile.gettempdir(), "kasa_batch_probe.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "batch_probe.c"), "-ldl"])
    return so


def run_probed(args, cwd, n_taxa, probe, keep=""):
    """Run the reference and return the number of reads in each of its batches (see batch_probe.c)."""
    tmp = os.path.join(cwd, "tmp")
    os.makedirs(tmp, exist_ok=True)
    log = os.path.join(tmp, "probe.txt")
    env = dict(os.environ, LD_PRELOAD=probe, KASA_PROBE_BYTES=str(4 * n_taxa), KASA_PROBE_OUT=log, KASA_PROBE_KEEP=keep)
    p = subprocess.run(KASA + args + ["-t", tmp + "/"], cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    if p.returncode != 0:
        sys.stderr.write(p.stdout)
        raise SystemExit("reference failed: " + " ".join(args))
    return [int(x) for x in open(log).read().split()]


def check_no_crowded_prefix(d):
    """Ground truth from the index the reference built: no scorable 7-letter prefix with more than 3 distinct taxa."""
    n = int(open(os.path.join(d, "idx_info.txt")).read().split()[0])
    pairs = set()
    with open(os.path.join(d, "idx"), "rb") as f:
        raw = f.read(n * 12)
    for i in range(n):
        kmer = int.from_bytes(raw[i * 12:i * 12 + 8], "little")
        if any(((kmer >> (5 * (11 - j))) & 31) == 30 for j in range(7)):
            continue                                           # '^' (genome tail): a query never matches past it
        pairs.add((kmer >> 25, raw[i * 12 + 8:i * 12 + 12]))
    per = {}
    for pre, tax in pairs:
        per[pre] = per.get(pre, 0) + 1
    worst = max(per.values())
    if worst > 3:
        raise SystemExit("index has a 7-letter prefix shared by %d taxa" % worst)


def gz(path):
    with open(path, "rb") as f, gzip.GzipFile(path + ".gz", "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    os.remove(path)


def case_batches(out):
    """An input the reference cuts into several batches (row A11): the -m budget is consumed mostly by the reads x taxa
    score matrix (Read.hpp:1192), so 100 000 placeholder taxa in the content file make a batch ~2 300 reads long and
    6 000 reads span three batches.  12 taxa x 10 kb with 2 % read errors leave enough unmatched stretches that the flush
    order of some reads really depends on their batch.  The per-read files differ between the budgets
    in the last float digit of some reads, which is what the multi-batch tests need to see.  batches.json records the
    batch sizes the binary really used (observed with batch_probe.c)."""
    rng = random.Random(11)
    G, L, DUMMY, NR = 12, 10000, 100000, 6000
    genomes = []
    for g in range(G):
        genomes.append(mutate(genomes[g - 1], 0.03, rng) if g % 2 else "".join(rng.choice("ACGT") for _ in range(L)))
    # The shipped binary is an AVX build: groups with more than 3 taxa take scoreMatchAVX, which drops increments
    # (Compare.hpp:534-597, the `clones` case).  Sibling pairs give 1-2 taxa per k-mer; where two pairs collide by chance
    # on a 7-letter prefix, one base is changed until no (k >= 7)-prefix of the index is shared by more than 3 taxa, so
    # this case runs the binary's scalar branch throughout -- the branch the device implements.
    for _ in range(50):
        seen = {}
        for g, s in enumerate(genomes):
            for fr in range(3):
                aa = translate(s[fr:])
                for i in range(len(aa) - 6):
                    seen.setdefault(aa[i:i + 7], {}).setdefault(g, fr + 3 * i)
        crowded = [v for v in seen.values() if len(v) > 3]
        if not crowded:
            break
        for v in crowded:
            g = max(v)
            pos = v[g] + 10                                    # inside the 7-letter window
            s = genomes[g]
            genomes[g] = s[:pos] + {"A": "C", "C": "G", "G": "T", "T": "A"}[s[pos]] + s[pos + 1:]
    else:
        raise SystemExit("could not thin out the crowded prefixes")
    write_db(out, genomes)
    with open(os.path.join(out, "content.txt"), "a") as c:
        for d in range(DUMMY):
            c.write("Placeholder %d\t%d\t%d\tDUM%06d.1\n" % (d, 1000 + d, 1000 + d, d))
    rng = random.Random(12)
    with open(os.path.join(out, "reads.fastq"), "w") as f:
        for r in range(NR):
            g = rng.randrange(G)
            p = rng.randrange(L - 150)
            s = mutate(genomes[g][p:p + 150], 0.02, rng)
            f.write("@read%d_t%d\n%s\n+\n%s\n" % (r, g, s, "I" * 150))
    run(["build", "-c", "content.txt", "-d", "idx", "-i", "db.fasta", "-m", "4", "-n", "1"], out)
    check_no_crowded_prefix(out)
    probe = build_probe()
    base = ["identify", "-c", "content.txt", "-d", "idx", "-n", "1", "-i", "reads.fastq", "--jsonl", "-b", "100"]
    sizes = {}
    for name, extra in (("m1", ["-m", "1"]), ("m2", ["-m", "2"]), ("m1_ram", ["-m", "1", "-r"]), ("m1_six", ["-m", "1", "--six"]),
                        ("m1_coh", ["-m", "1", "--coherence"])):
        sizes[name] = run_probed(base + extra + ["-q", "out_%s.jsonl" % name, "-p", "prof_%s.csv" % name], out, G + DUMMY + 1, probe)
        assert sum(sizes[name]) == NR and len(sizes[name]) >= 3, sizes
    with open(os.path.join(out, "batches.json"), "w") as f:
        json.dump(sizes, f, indent=1)
    for big in ["content.txt", "idx_f.txt", "reads.fastq", "db.fasta"] + ["out_%s.jsonl" % n for n in sizes]:
        gz(os.path.join(out, big))


def case_longseq(out):
    """A sequence the reference cuts into pieces (Read.hpp:372-467: a piece ends where its k-mers pass 100 MiB) and
    carries across two batches (strTransfer, Read.hpp:343-356; the pieces' scores merged in Compare.hpp:2344-2426).
    The database, content file and index are the `batches` case's (rebuilt here from its committed inputs and checked
    against its committed index); only the input and the reference's outputs are new, and they are stored next to it:
    short reads that use up most of the -m 1 budget, then one 9.5 Mbp sequence made of mutated copies of the database's
    genomes (three pieces: the first ends batch 1, the other two share a read id in batch 2), then more short reads."""
    src = os.path.join(HERE, "batches")
    for name in ("db.fasta", "content.txt"):
        with gzip.open(os.path.join(src, name + ".gz"), "rb") as f, open(os.path.join(out, name), "wb") as g:
            shutil.copyfileobj(f, g)
    genomes, cur = [], []
    for line in open(os.path.join(out, "db.fasta")):
        if line.startswith(">"):
            if cur:
                genomes.append("".join(cur))
            cur = []
        else:
            cur.append(line.strip())
    genomes.append("".join(cur))
    G, L = len(genomes), len(genomes[0])
    n_taxa = sum(1 for _ in open(os.path.join(out, "content.txt"))) + 1
    run(["build", "-c", "content.txt", "-d", "idx", "-i", "db.fasta", "-m", "4", "-n", "1"], out)
    n = int(open(os.path.join(out, "idx_info.txt")).read().split()[0])
    with open(os.path.join(out, "idx"), "rb") as f, open(os.path.join(src, "idx"), "rb") as g:
        assert f.read(n * 12) == g.read(), "the rebuilt index differs from tests/golden/batches/idx"
    # a repeat unit none of whose k-mers (any frame, either strand) shares a 7-letter prefix with the index: most of sequence B
    # is made of it, so that the device (which replays the events of ONE read with one wavefront)
    # and the oracle get through its 24.5 Mbp quickly -- what the case is about are the pieces and the batches
    seen7 = set()
    for s_ in genomes:
        for fr in range(3):
            aa = translate(s_[fr:])
            seen7.update(aa[i:i + 7] for i in range(len(aa) - 6))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    urng = random.Random(44)
    while True:
        unit = "".join(urng.choice("ACGT") for _ in range(11))
        rep = unit * 12
        both = [rep, "".join(comp[c] for c in reversed(rep))]
        if not any(translate(x[fr:])[i:i + 7] in seen7 for x in both for fr in range(3) for i in range(30)):
            break

    def write_input(path, rng, layout):
        count = [0]
        with open(path, "w") as f:
            for item in layout:
                if item[0] == "short":
                    for _ in range(item[1]):
                        g = rng.randrange(G)
                        p = rng.randrange(L - 150)
                        f.write(">s%d_t%d\n%s\n" % (count[0], g, mutate(genomes[g][p:p + 150], 0.02, rng)))
                        count[0] += 1
                    continue
                _, title, left, rate = item[:4]
                filler = item[4] if len(item) > 4 else 0       # letters of `unit` (matches nothing in the index) behind every genome copy
                f.write(">%s\n" % title)
                count[0] += 1
                col = 0
                while left > 0:
                    s = mutate(genomes[rng.randrange(G)], rate, rng)
                    if filler:
                        s += (unit * (filler // len(unit) + 1))[:filler]
                    s = s[:left]
                    left -= len(s)
                    i = 0
                    while i < len(s):                              # 70 letters per line
                        take = min(70 - col, len(s) - i)
                        f.write(s[i:i + take])
                        i += take
                        col += take
                        if col == 70:
                            f.write("\n")
                            col = 0
                if col:
                    f.write("\n")

    write_input(os.path.join(out, "long.fasta"), random.Random(41),
                [("short", int(os.environ.get("KASA_LONGSEQ_SHORT", "2100"))), ("contig", "contig made of the database", 9500000, 0.003), ("short", 40)])
    # ... and an input whose batches end inside sequences in every way there is (six frames: pieces of 2.2 Mbp): sequence A in
    # two pieces across batches 1 and 2; batch 2 goes on with short reads and ends after the first piece of sequence B (a
    # batch that begins AND ends inside a sequence); batch 3 lies inside B altogether; batch 4 finishes it
    S1, S2 = int(os.environ.get("KASA_LONGSEQ_S1", "2150")), int(os.environ.get("KASA_LONGSEQ_S2", "1900"))
    write_input(os.path.join(out, "long2.fasta"), random.Random(43),
                [("short", S1), ("contig", "sequence A", 3000000, 0.001), ("short", S2), ("contig", "sequence B", 24500000, 0.0005, 190000), ("short", 25)])
    # ... and the first input once more as FASTQ, the long sequence and its quality string on ONE line each: the reader's
    # pieces end where its 2048-byte buffers end, line feeds are not counted (Read.hpp:469-598)
    with open(os.path.join(out, "long.fasta")) as f, open(os.path.join(out, "long3.fastq"), "w") as g:
        name, seq = None, []
        for line in list(f) + [">"]:
            if line.startswith(">"):
                if name is not None:
                    g.write("@%s\n%s\n+\n%s\n" % (name, "".join(seq), "I" * sum(len(x) for x in seq)))
                name, seq = line[1:].rstrip("\n"), []
            else:
                seq.append(line.rstrip("\n"))
    probe = build_probe()
    sizes = {}
    for name, stem, extra in (("long", "long.fasta", []), ("long_six", "long.fasta", ["--six"]), ("long2_six", "long2.fasta", ["--six"]), ("long3", "long3.fastq", [])):
        batches = run_probed(["identify", "-c", "content.txt", "-d", "idx", "-n", "1", "-i", stem, "--jsonl", "-b", "100", "-m", "1"] + extra +
                             ["-q", "out_%s.jsonl" % name, "-p", "prof_%s.csv" % name], out, n_taxa, probe, keep="_fileInfo.txt")
        # the reference's own list of pieces (skip lines, getChunk calls, pieces left), kept from deletion by the probe: the
        # lines of the long sequence are what the restatement of Read.hpp:372-467 has to reproduce
        info = os.path.join(out, "tmp", stem.rsplit(".", 1)[0] + "_fileInfo.txt")
        pieces = [l.strip() for l in open(info) if int(l.split(",")[2]) > 1 or (l.startswith("0,") and not l.strip().endswith(",0"))]
        os.remove(info)
        sizes[name] = {"batches": batches, "pieces_of_the_long_sequence": pieces}
    # ... and the first input with --filter and the TSV writer (-b 3): the read that is finished from what a batch before
    # left of it goes through the reference's one-read writer (Compare.hpp:1894-2265) and its own filter test
    import hashlib
    run_probed(["identify", "-c", "content.txt", "-d", "idx", "-n", "1", "-i", "long.fasta", "--tsv", "-b", "3", "-m", "1", "--filter", "lflt_clean", "lflt_cont",
                "-q", "out_long_flt.tsv", "-p", "prof_long_flt.csv"], out, n_taxa, probe)
    sizes["long_flt"] = {"sha256": {n: hashlib.sha256(open(os.path.join(out, n), "rb").read()).hexdigest() for n in ("lflt_clean.fasta", "lflt_cont.fasta")},
                         "reads": {n: sum(1 for l in open(os.path.join(out, n)) if l.startswith(">")) for n in ("lflt_clean.fasta", "lflt_cont.fasta")}}
    with open(os.path.join(out, "out_long_flt.tsv"), "rb") as f, gzip.GzipFile(os.path.join(src, "out_long_flt.tsv.gz"), "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    with open(os.path.join(src, "long.json"), "w") as f:
        json.dump(sizes, f, indent=1)
    for stem in ("long.fasta", "long2.fasta", "long3.fastq"):
        with open(os.path.join(out, stem), "rb") as f, lzma.open(os.path.join(src, stem + ".xz"), "wb", preset=9) as g:
            shutil.copyfileobj(f, g)
    for name in sizes:
        if name == "long_flt":
            continue
        with open(os.path.join(out, "out_%s.jsonl" % name), "rb") as f, gzip.GzipFile(os.path.join(src, "out_%s.jsonl.gz" % name), "wb", mtime=0) as g:
            shutil.copyfileobj(f, g)
        shutil.copy(os.path.join(out, "prof_%s.csv" % name), os.path.join(src, "prof_%s.csv" % name))


def finish(out):
    trim_index(out)
    for junk in ("tmp", "stxxl.log", "stxxl.errlog"):
        p = os.path.join(out, junk)
        if os.path.isdir(p):
            shutil.rmtree(p)
        elif os.path.exists(p):
            os.remove(p)


def main():
    if not os.path.exists(KASA[1]):
        raise SystemExit("needs /root/reference (development container only)")
    only = sys.argv[1:]
    for name, fn in (("pairs", case_pairs), ("clones", case_clones), ("batches", case_batches)):
        if only and name not in only:
            continue
        out = os.path.join(HERE, name)
        if os.path.isdir(out):
            shutil.rmtree(out)
        os.makedirs(out)
        fn(out)
        finish(out)
        print("wrote", out, sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)) // 1024, "KiB")
    if not only or "longseq" in only or "batches" in only:      # lives in batches/ (same database): goes when that is rebuilt
        keep = os.environ.get("KASA_LONGSEQ_DIR")
        with tempfile.TemporaryDirectory() as tmp:
            work = keep or tmp
            os.makedirs(work, exist_ok=True)
            case_longseq(work)
        print("wrote the long-sequence case into", os.path.join(HERE, "batches"))
    ver = subprocess.run(KASA, stdout=subprocess.PIPE, text=True, timeout=60).stdout.splitlines()[0]
    with open(os.path.join(HERE, "PROVENANCE.json"), "w") as f:
        json.dump({"reference_binary": "binaries/kASA_linux", "banner": ver.split(" ran on")[0],
                   "note": "AVX build: groups with more than 3 taxa go through scoreMatchAVX"}, f, indent=1)


if __name__ == "__main__":
    main()
