"""CPU-only checks: the C-ABI library loads and exports every symbol of include/kasa_hip.h, number
text, file formats, read parsing.  No compute call is made here (no GPU in this tier)."""
import os
import re

import numpy as np
import pytest

from kasa_amd import capi, formats, reads, textnum
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "kasa_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|uint64_t|int64_t|const char \*)\s*\*?(kasa_[a-z_0-9]+)\(", header, re.M))
    L = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(capi.EXPORTS)


def test_no_cpu_fallback_without_gpu():
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    d, ix = helpers.load_case("pairs")
    with pytest.raises(RuntimeError):
        capi.DeviceIndex(ix)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under kasa_amd/ may import, include, link or call it."""
    bad = re.compile(r"^\s*(?:import|from)\s+oracle\b|#include[^\n]*oracle|libkasa_oracle|\bko_[a-z_]+\s*\(", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kasa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not bad.search(src), f


def test_cached_powers_match_published_constants():
    # first / second / last entries of the Grisu cached-power table (Loitsch 2010; double-conversion)
    assert textnum._POWERS[0] == (0xfa8fd5a0081c0288, -1220)
    assert textnum._POWERS[1] == (0xbaaee17fa23ebf76, -1193)
    assert textnum._POWERS[44] == (0x9c40000000000000, -50)          # 10^4, exact
    assert textnum._POWERS[86][1] == 1066


def test_dtoa_reproduces_every_number_the_reference_printed(golden_dir):
    pat = re.compile(r"(?<![\w.])-?\d+\.\d+(?:e-?\d+)?(?![\w.])")
    n = 0
    for case in ("pairs", "clones"):
        d = os.path.join(golden_dir, case)
        for f in sorted(os.listdir(d)):
            if f.startswith("out_") and not f.endswith(".ktsv"):
                for tok in pat.findall(open(os.path.join(d, f), errors="replace").read()):
                    assert textnum.dtoa(float(tok)) == tok, (f, tok)
                    n += 1
    assert n > 2000


def test_dtoa_roundtrip_random():
    rng = np.random.default_rng(3)
    for v in rng.random(3000) * 200:
        assert float(textnum.dtoa(float(v))) == float(v)
    for v in rng.random(3000).astype(np.float32) * np.float32(100):
        assert float(textnum.dtoa(float(v))) == float(v)
    assert textnum.dtoa(0.0) == "0.0" and textnum.dtoa(1e21) == "1e21" and textnum.dtoa(100.0) == "100.0"


def test_index_files_roundtrip(tmp_path, golden_dir):
    d, ix = helpers.load_case("pairs")
    assert ix.n == int(open(os.path.join(d, "idx_info.txt")).read())
    assert (np.diff(ix.kmer.astype(np.float64)) >= 0).all()
    # our trie/frequency derivation equals what the reference's build wrote
    tp, tc = formats.trie_from_kmers(ix.kmer)
    assert np.array_equal(tp, ix.trie_prefix) and np.array_equal(tc, ix.trie_count)
    assert np.array_equal(formats.freq_from_index(ix.kmer, ix.tax, ix.content.n_taxa), ix.freq)
    formats.write_index(ix, str(tmp_path / "idx"), str(tmp_path / "content.txt"))
    ix2 = formats.load_index(str(tmp_path / "idx"), str(tmp_path / "content.txt"))
    assert np.array_equal(ix2.kmer, ix.kmer) and np.array_equal(ix2.tax, ix.tax)
    assert np.array_equal(ix2.freq, ix.freq)


def test_read_parsing_matches_reference_conventions(golden_dir):
    d = os.path.join(golden_dir, "pairs")
    fq = reads.parse_reads(os.path.join(d, "reads.fastq"))
    fa = reads.parse_reads(os.path.join(d, "reads.fasta"))
    assert fq.n == fa.n and fq.names == fa.names
    assert np.array_equal(fq.bases, fa.bases)
    assert fq.names[0].endswith(" ")
    assert int(fq.lengths[0]) == 151                     # one sequence line: +1
    assert int(fa.lengths[0]) == 150 + 3                 # three 60-column lines: +3
    with pytest.raises(RuntimeError):
        p = os.path.join(str(d), "..", "PROVENANCE.json")
        reads.parse_reads(p)


def test_halved_index_writer_matches_reference_shrink(tmp_path):
    """formats.write_index_halved == the files `kASA shrink -s 2` wrote for the same index (tests/golden/pairs/idx_half*)."""
    import os
    from kasa_amd import formats
    from tests import helpers
    d = os.path.join(helpers.GOLDEN, "pairs")
    ix = formats.load_index(os.path.join(d, "idx"), os.path.join(d, "content.txt"))
    out = str(tmp_path / "h")
    formats.write_index_halved(ix, out)
    for suffix in ("", "_trie", "_trie.txt", "_info.txt", "_f.txt"):
        with open(out + suffix, "rb") as a, open(os.path.join(d, "idx_half" + suffix), "rb") as b:
            assert a.read() == b.read(), suffix


def _build_stdsort_check(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "stdsort_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "stdsort_check.cpp")], check=True)
    return exe


def test_stdsort_order_equals_libstdcxx(tmp_path):
    """kasa_amd/csrc/stdsort_order.h (the order std::sort leaves tied hits in, restated for the device ranking) against
    std::sort itself: 60 000 arrays of 1..1500 elements with many, some and no ties, sorted and reversed inputs."""
    import subprocess
    r = subprocess.run([_build_stdsort_check(tmp_path)], stdout=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK "), r.stdout


def test_python_stdsort_order_equals_libstdcxx(tmp_path):
    """... and the Python restatement (kasa_amd/report.py:_stdsort_order, used by rank_read for more than 16 hits) against
    std::sort's order on 300 tie-heavy arrays printed by the same program."""
    import subprocess
    from kasa_amd import report
    r = subprocess.run([_build_stdsort_check(tmp_path), "dump"], stdout=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0
    lines = r.stdout.strip().split("\n")
    assert len(lines) == 300
    for line in lines:
        rels, ids = line.split("|")
        rel = [float(x) for x in rels.split()]
        want = [int(x) for x in ids.split()]
        got = report._stdsort_order(len(rel), lambda x, y: rel[x] > rel[y])
        assert got == want


@pytest.mark.parametrize("name", ["reads.fastq", "reads.fasta", "edge_crlf.fasta", "edge_multi.fastq", "edge_noeol.fasta", "exampleInput.fasta",
                                  "reads_prot.fasta", "reads_dup.fastq"])
@pytest.mark.parametrize("block,run", [(None, None), ("1500", "400"), ("97", "33")])
def test_cpp_parser_equals_python_parser(name, block, run, golden_dir):
    """The C++ driver's streaming parser (blocks that take turns, runs parsed by several threads, huge-page arrays) against
    kasa_amd/reads.py on the reference's own kinds of input -- without a device: `kasa_identify parse-dump` is a test tap."""
    import subprocess
    from kasa_amd import build as hipbuild, reads
    exe = hipbuild.HOST_BIN
    if not os.path.exists(exe):
        pytest.skip("host driver not built (python -m kasa_amd.build)")
    path = os.path.join(golden_dir, "pairs", name)
    env = dict(os.environ)
    if block:
        env.update(KASA_READ_BLOCK=block, KASA_PARSE_CHUNK=run)
    r = subprocess.run([exe, "parse-dump", path, "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-400:]
    out = r.stdout.decode("latin-1")
    whole, streamed = out.split("== streamed\n")
    ref = reads.parse_reads(path)
    want = "".join(f"{ref.names[i]}\t{int(ref.lengths[i])}\t{bytes(ref.bases[int(ref.offsets[i]):int(ref.offsets[i + 1])]).decode('latin-1')}\n" for i in range(ref.n))
    for part in (whole.split("== whole file\n")[1], streamed):
        head, body = part.split("\n", 1)
        assert head.startswith("protein=%d" % (1 if ref.protein else 0)), head
        assert body == want


def test_crowded_genomes_share_what_they_claim():
    """synth.genomes_crowded (bench.py's `tertiary`, the crowded id of the full-size test): members of a clade agree on the
    clade's gene intervals up to their 1-5 % of substitutions, every taxon carries the universal intervals, the rest is its own."""
    from kasa_amd import synth
    g = synth.genomes_crowded(40, 30_000, seed=3, clade_sizes=(10, 20))
    assert g.shape == (40, 30_000) and set(np.unique(g)) <= set(b"ACGT")
    same01 = float((g[0] == g[1]).mean())                       # two members of the first clade: 30 % + 5 % shared (at ~94 % identity) + chance
    assert 0.42 < same01 < 0.54, same01
    far = float((g[0] == g[39]).mean())                         # another clade: only the universal 5 % + chance (25 % of the rest)
    assert 0.26 < far < 0.34, far
    g2 = synth.genomes_crowded(40, 30_000, seed=3, clade_sizes=(10, 20))
    assert np.array_equal(g, g2)                                # seeded



def _run_py(code):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % root + code], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, KASA_QUIET_HIP_RUNTIME="1"))


def _lib_and_torch_or_skip():
    import importlib.util
    from kasa_amd import capi
    if not os.path.exists(capi.SO_PATH) or importlib.util.find_spec("torch") is None:
        pytest.skip("libkasa_hip.so or torch is not here")


def test_one_hip_runtime_when_torch_comes_after_the_library():
    """A torch wheel carries its own libamdhip64.so.  Imported after libkasa_hip.so had pulled in ROCm's copy it is a second
    HIP runtime in the process and torch finds no device.  A host that will import torch says so first
    (capi.share_torch_runtime()): then there is ONE runtime, whatever the order -- and it is an explicit choice (round 4 made
    the swap for every process, torch or not)."""
    _lib_and_torch_or_skip()
    r = _run_py("from kasa_amd import capi\n"
                "capi.share_torch_runtime()\n"
                "capi.lib()\n"
                "import torch\n"
                "info = capi.runtime_info()\n"
                "print(len(info['mapped']['libamdhip64']), info['runtime_from'])\n")
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.split()[0] == "1" and "torch" in r.stdout, r.stdout


def test_own_hip_runtime_without_torch():
    """A process that never asks for torch runs on the runtime the library was built for (ROCm's own): versions equal."""
    _lib_and_torch_or_skip()
    r = _run_py("from kasa_amd import capi\n"
                "info = capi.runtime_info()\n"
                "import sys\n"
                "assert 'torch' not in sys.modules\n"
                "print(info['hip_built'], info['hip_runtime'], info['hip_runtime_matches_build'], info['mapped']['libamdhip64'])\n")
    assert r.returncode == 0, r.stderr[-500:]
    assert "True" in r.stdout and "/opt/rocm" in r.stdout, r.stdout


def test_torch_first_is_one_runtime_and_a_mismatch_is_told():
    """torch imported first: its runtime is the process's; the library announces (or, strict, refuses) a build/runtime
    mismatch instead of running on it unawares."""
    _lib_and_torch_or_skip()
    code = ("import torch\n"
            "from kasa_amd import capi\n"
            "info = capi.runtime_info()\n"
            "print(len(info['mapped']['libamdhip64']), info['hip_runtime_matches_build'])\n")
    r = _run_py(code)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.split()[0] == "1"
    if r.stdout.split()[1] == "False":
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r2 = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % root + code], capture_output=True, text=True, timeout=300,
                            env=dict(os.environ, KASA_STRICT_HIP_RUNTIME="1"))
        assert r2.returncode != 0 and "KASA_STRICT_HIP_RUNTIME" in r2.stderr


def test_library_does_not_link_rccl():
    """RCCL is bound at run time to the copy the process holds (kasa_profile_allreduce): the library itself must not pull a
    second one in."""
    import subprocess
    from kasa_amd import capi
    if not os.path.exists(capi.SO_PATH):
        pytest.skip("libkasa_hip.so is not built")
    out = subprocess.run(["readelf", "-d", capi.SO_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in out and "libamdhip64.so" in out


def test_wire_format_of_exported_records_round_trips():
    """partition.pack_records / unpack_records (the numpy statement of kasa_batch_records_pack's wire format): random records of
    both widths come back with word [0] = their place and their unused words zero; an unmatched record costs two bits."""
    import numpy as np
    from kasa_amd import partition
    rng = np.random.default_rng(5)
    for rw in (8, 16):
        for n in (0, 1, 3, 4, 5, 63, 64, 65, 1003):
            rec = rng.integers(0, 2 ** 32, size=(n, rw), dtype=np.uint64).astype(np.uint32)
            rec[:, 0] = np.arange(n, dtype=np.uint32)
            d = rng.integers(0, 13, n).astype(np.uint32)
            rec[:, 2] = (rec[:, 2] & ~np.uint32(31)) | d
            ns = rng.integers(0, 12, n).astype(np.uint32)
            rec[:, 3] = ((rec[:, 3] & ~np.uint32(255)) | ns) if rw == 8 else ns
            wire = partition.pack_records(rec, rw)
            back = partition.unpack_records(wire, n, rw)
            nw = partition._WIRE_WORDS[rw][partition._wire_classes(rec, rw)] if n else np.zeros(0, dtype=np.int64)
            want = rec.copy()
            for i in range(n):
                want[i, 1 + nw[i]:] = 0
            assert np.array_equal(back, want)
            assert wire.nbytes == ((n + 3) // 4 + 15) // 16 * 16 + 4 * int(nw.sum())
    none = np.zeros((100, 8), dtype=np.uint32)
    assert partition.pack_records(none, 8).nbytes == 32
