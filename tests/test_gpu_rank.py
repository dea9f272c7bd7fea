"""Ranking on the device (kasa_batch_rank): only the hits the per-read file can print cross
PCIe.  The text written from them must be the text the host writes after ranking the full rows itself
(report.rank_read = Compare::scoringFunc, Compare.hpp:1495-1594, 1721-1754) -- for every output format, -b, threshold,
mixed read lengths, and for the reads the device hands back (ties among more than 16 hits)."""
import os
import subprocess

import numpy as np
import pytest

from kasa_amd import build as hipbuild, capi, formats, identify, reads, report
from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu


def _mixed_lengths(genomes, seed):
    parts = [reads.synthetic_reads(genomes, n, L, seed + i) for i, (n, L) in enumerate([(300, 150), (200, 100), (100, 251), (50, 40)])]
    bases = np.concatenate([p.bases for p in parts])
    off, names, lens, at = [0], [], [], 0
    for p in parts:
        for r in range(p.n):
            at += int(p.offsets[r + 1] - p.offsets[r])
            off.append(at)
            names.append(f"m{len(names)} ")
            lens.append(int(p.lengths[r]))
    return reads.ReadBatch(bases, np.asarray(off, dtype=np.int64), names, np.asarray(lens, dtype=np.uint32))


def _world(kind):
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    if kind == "crowd":                       # 300 taxa around one root: dozens of hits per read
        root = alphabet[rng.integers(0, 4, size=900)]
        genomes = []
        for g in range(300):
            s = root.copy()
            m = rng.random(900) < 0.1
            s[m] = alphabet[rng.integers(0, 4, size=int(m.sum()))]
            genomes.append(s)
    if kind == "clones":                      # 40 identical genomes: equal k-mer scores, equal frequencies, equal relative scores
        one = alphabet[rng.integers(0, 4, size=3000)]
        genomes = [one.copy() for _ in range(40)]
    if kind == "pairs":                       # sibling genomes: two or three close hits per read
        rs = np.random.default_rng(23)
        genomes = []
        for g in range(12):
            if g % 2 == 1:
                s = genomes[g - 1].copy()
                m = rs.random(6000) < 0.03
                s[m] = alphabet[rs.integers(0, 4, size=int(m.sum()))]
            else:
                s = alphabet[rs.integers(0, 4, size=6000)]
            genomes.append(s)
    content = formats.Content(["non_unique"] + [f"Taxon, {g}" for g in range(len(genomes))],
                              np.concatenate(([0], 100 + np.arange(len(genomes)))).astype(np.uint32))
    p = oracle.params(12, 7, 3)
    kms, tids = [], []
    for g, s in enumerate(genomes):
        km, _ = oracle.encode(s, np.array([0, s.shape[0]], dtype=np.int64), p)
        kms.append(km)
        tids.append(np.full(km.shape[0], 100 + g, dtype=np.uint32))
    ix = formats.make_index(np.concatenate(kms), np.concatenate(tids), content)
    return ix, _mixed_lengths(genomes, 40)


@pytest.mark.parametrize("kind", ["pairs", "crowd", "clones"])
def test_device_rank_writes_the_hosts_text(kind):
    assert capi.device_count() > 0
    ix, batch = _world(kind)
    dix = capi.DeviceIndex(ix)
    flagged = handed_back = 0
    for fmt in ("json", "jsonl", "tsv", "kraken"):
        for beasts, thr in ((1, 0.0), (3, 0.0), (7, 0.03), (100, 0.0)):
            texts = []
            for mode in ("device", "device rank, host text", "device, ties to the host", "host"):
                run = identify.Identify(ix, 0, 12, 7, 3, thr, beasts, fmt, dix=dix)
                run.device_rank = mode != "host"
                run.device_text = mode == "device"
                run.ctx.debug_flags(256 if mode == "device, ties to the host" else 0)
                text, prof, _ = run.run(batch, True)
                texts.append((text, prof))
                if mode == "device":
                    flagged += run.flagged_reads
                    assert run.device_text_batches == 1            # the bytes came from kasa_batch_text
                else:
                    assert run.device_text_batches == 0
                if mode == "device, ties to the host":
                    handed_back += run.flagged_reads
                run.close()
            assert texts[0] == texts[1] == texts[2] == texts[3], (fmt, beasts, thr)
    assert flagged == 0                       # ties among more than 16 hits take std::sort's own order on the device
    if kind == "clones":
        assert handed_back > 0                # 40 tied hits per read: with the test tap those reads go back to the host
    dix.close()


def test_rank_entries_against_rank_read():
    """The raw ABI: entries = a prefix of the host's sorted hits, meta = (first, count | flag, max k-mer score, hits)."""
    ix, batch = _world("crowd")
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(batch.bases, batch.offsets, True)
    off, tax, sc = ctx.scores()
    freq = ix.freq_at(12)
    den, rclass = report.rank_denominators(freq, batch.lengths, ix.K, False)
    for beasts in (0, 1, 3, 50):
        meta, ent, n_flag = ctx.rank(den, rclass, 0.0, beasts, pinned=(beasts != 1))
        assert meta.shape == (batch.n, 4)
        for r in range(batch.n):
            lo, hi = int(off[r]), int(off[r + 1])
            rk = report.rank_read(tax[lo:hi], sc[lo:hi], int(batch.lengths[r]), freq, 12, 7, 3, 0.0, beasts, K=ix.K)
            assert int(meta[r, 3]) == len(rk.hits)
            if int(meta[r, 1]) >> 31:
                continue
            a, n = int(meta[r, 0]), int(meta[r, 1])
            assert n <= len(rk.hits) and (n >= min(len(rk.hits), max(beasts, 1)))
            got = [(int(e["tax"]), float(e["score"]), float(e["rel"])) for e in ent[a:a + n]]
            want = [(h.tax_idx, float(h.score), h.rel) for h in rk.hits[:n]]
            assert got == want                # bit-equal doubles: IEEE division on both sides, libm's log2 from the host
            if rk.hits:
                assert np.uint32(meta[r, 2]).view(np.float32) == max(h.score for h in rk.hits)
    ctx.close(); dix.close()


def test_cpp_host_device_rank_equals_host_rank(tmp_path):
    exe = hipbuild.build_host()
    d = os.path.join(helpers.GOLDEN, "pairs")
    outs = []
    for extra in ([], ["--host-rank"]):
        out, prof = str(tmp_path / ("out" + str(len(outs)))), str(tmp_path / "prof.csv")
        cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq"),
               "-q", out, "-p", prof, "--json", "-b", "5", "-n", "2"] + extra
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 1000


@pytest.mark.parametrize("kind", ["clones", "crowd"])
def test_cpp_host_ties_follow_std_sort(kind, tmp_path):
    """The C++ driver ranks with the real std::sort under --host-rank and with the device's restatement of it by default:
    on inputs full of tied hits (40 identical genomes; 300 taxa around one root) the files must be the same bytes."""
    exe = hipbuild.build_host()
    ix, batch = _world(kind)
    formats.write_index(ix, str(tmp_path / "idx"), str(tmp_path / "content.txt"))
    with open(tmp_path / "reads.fastq", "wb") as f:
        for r in range(batch.n):
            seq = batch.bases[int(batch.offsets[r]):int(batch.offsets[r + 1])].tobytes()
            f.write(b"@" + batch.names[r].rstrip().encode() + b"\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n")
    for fmt, flag in (("jsonl", "--jsonl"), ("tsv", "--tsv")):
        outs = []
        for extra in ([], ["--host-rank"]):
            out = str(tmp_path / ("out_%s_%d" % (fmt, len(outs))))
            cmd = [exe, "identify", "-c", str(tmp_path / "content.txt"), "-d", str(tmp_path / "idx"), "-i", str(tmp_path / "reads.fastq"),
                   "-q", out, "-p", str(tmp_path / "prof.csv"), flag, "-b", "5", "-n", "2"] + extra
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            assert r.returncode == 0, r.stderr
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] and len(outs[0]) > 1000, fmt


def test_device_number_format():
    """kasa_text.h's Grisu2 against the host's (kasa_amd/textnum.py, pinned on the reference's files): doubles of every
    magnitude, float32 values widened (what k-mer scores and errors are), exact powers, subnormals, the specials."""
    from kasa_amd import textnum
    rs = np.random.default_rng(5)
    vals = [0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 1e21, 1e22, 1e-5, 1e-6, 1e-7, 123456789012345678.0, 5e-324, 2.2250738585072014e-308,
            1.7976931348623157e308, float("inf"), float("-inf"), float("nan"), 0.3, 2.0 / 3.0, 100.0, 1e20, 9.999999999999999e20]
    vals += [10.0 ** k for k in range(-30, 31)]
    vals += list(rs.random(3000))
    vals += list((rs.random(3000) * 2 - 1) * 10.0 ** rs.integers(-12, 25, size=3000))
    vals += [float(x) for x in rs.random(3000).astype(np.float32)]                                # float -> double as the writer passes them
    vals += [float(np.float32(a) / np.float32(b)) for a, b in zip(rs.integers(1, 5000, 2000), rs.integers(1, 5000, 2000))]
    vals += list(np.frombuffer(rs.bytes(8 * 3000), dtype=np.float64))                           # any bit pattern
    got = capi.device_dtoa(vals)
    for v, g in zip(vals, got):
        assert g == textnum.dtoa(float(v)), (v, g, textnum.dtoa(float(v)))


def test_device_text_with_awkward_names():
    """Specifiers and taxon names are printed as they are (no escaping in the reference): empty names, quotes, backslashes,
    bytes beyond ASCII, a name of a kilobyte (beyond any short-string buffer); beasts 1..5; with and without a threshold."""
    ix, batch = _world("pairs")
    odd = ['', '"quoted"', 'back\\slash', 'caf\xe9 \xfc\xf1', 'x' * 1000, 'a;b,c', "tab\there", ' ', '{', '}']
    ix.content.names = [ix.content.names[0]] + [odd[i % len(odd)] + (str(i) if i % 3 else "") for i in range(len(ix.content.names) - 1)]
    batch.names = [odd[(7 * i) % len(odd)] + ("" if i % 5 == 0 else f"r{i} ") for i in range(batch.n)]
    dix = capi.DeviceIndex(ix)
    for fmt in ("json", "jsonl", "tsv", "kraken"):
        for beasts, thr in ((1, 0.0), (2, 0.02), (5, 0.0)):
            texts = []
            for device_text in (True, False):
                run = identify.Identify(ix, 0, 12, 7, 3, thr, beasts, fmt, dix=dix)
                run.device_text = device_text
                text, prof, _ = run.run(batch, True)
                assert run.device_text_batches == (1 if device_text else 0)
                texts.append(text)
                run.close()
            assert texts[0] == texts[1], (fmt, beasts, thr)
    dix.close()


def test_text_abi_edges():
    """kasa_batch_text refuses what it cannot do; kasa_batch_text_fetch_range hands out any piece and nothing beyond the text;
    kasa_ctx_reserve before a batch changes nothing but where the buffers come from."""
    import ctypes as C
    ix, batch = _world("pairs")
    dix = capi.DeviceIndex(ix)
    lib = capi.lib()
    texts = []
    for reserve in (False, True):
        ctx = capi.Context(dix, 12, 7, 3)
        if reserve:
            assert lib.kasa_ctx_reserve(ctx.h, C.c_uint64(5_000_000), C.c_uint64(1_000_000), C.c_int(1)) == 0
        ctx.run_batch(batch.bases, batch.offsets, True)
        freq = ix.freq_at(12)
        den, rclass = report.rank_denominators(freq, batch.lengths, ix.K, False)
        best = np.array([report.best_score(int(L), 12, 7, 3, False) for L in np.unique(batch.lengths)], dtype=np.float32)
        with pytest.raises(RuntimeError):                           # not ranked yet
            ctx.text("jsonl", 3, 0, batch.names, batch.lengths, best)
        ctx.rank(den, rclass, 0.0, 3)
        with pytest.raises(RuntimeError):                           # no taxon names yet
            ctx.text("jsonl", 3, 0, batch.names, batch.lengths, best)
        ctx.set_taxa_text(ix.content.taxids, ix.content.names)
        whole, offs, _ = ctx.text("jsonl", 3, 7, batch.names, batch.lengths, best)
        pieces, _, _ = ctx.text("jsonl", 3, 7, batch.names, batch.lengths, best, pieces=5)
        assert whole == pieces and int(offs[-1]) == len(whole) and whole.count(b"\n") == batch.n
        assert whole.startswith(b'{ "Read number": 7, ')
        buf = np.zeros(16, dtype=np.uint8)
        assert lib.kasa_batch_text_fetch_range(ctx.h, C.c_void_p(buf.ctypes.data), C.c_uint64(len(whole) - 8), C.c_uint64(16)) != 0   # beyond the end
        assert lib.kasa_batch_text_fetch_range(ctx.h, C.c_void_p(buf.ctypes.data), C.c_uint64(len(whole) - 8), C.c_uint64(8)) == 0
        assert bytes(buf[:8]) == whole[-8:]
        texts.append(whole)
        ctx.close()
    assert texts[0] == texts[1]
    dix.close()
