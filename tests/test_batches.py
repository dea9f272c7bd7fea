"""Row A11: per-read files byte-identical to the reference ACROSS its batch boundaries.

tests/golden/batches holds an input the reference binary cut into three batches (make_fixtures.py:case_batches; the
batch sizes in batches.json were observed with batch_probe.c).  Per-read float sums depend on the reads that share a
batch, so these outputs can only be reproduced by a host that cuts where `kASA identify -m <GiB>` cuts:

* CPU: kasa_refbatch_* (host arithmetic of the C ABI) reproduces the observed batch sizes; the oracle run batch by batch
  reproduces the reference's files byte for byte -- and does NOT when the whole input is one batch (the fixture is
  sensitive to the boundaries);
* GPU: the Python host (memory_gib=) and the C++ driver (-m) write the same bytes, scores included.
"""
import gzip
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from kasa_amd import capi, formats, reads
from tests import helpers

SRC = os.path.join(helpers.GOLDEN, "batches")
CONFIGS = {  # name -> (-m, -r, frames)
    "m1": (1, False, 3), "m2": (2, False, 3), "m1_ram": (1, True, 3), "m1_six": (1, False, 6),
}


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("batches"))
    for f in os.listdir(SRC):
        src = os.path.join(SRC, f)
        if f.endswith(".gz") and not f.startswith("reads"):
            with gzip.open(src, "rb") as g, open(os.path.join(d, f[:-3]), "wb") as o:
                shutil.copyfileobj(g, o)
        else:
            shutil.copy(src, os.path.join(d, f))
    ix = formats.load_index(os.path.join(d, "idx"), os.path.join(d, "content.txt"))
    batch = reads.parse_reads(os.path.join(d, "reads.fastq.gz"))
    sizes = json.load(open(os.path.join(d, "batches.json")))
    return d, ix, batch, sizes


def _text(d, name):
    with open(os.path.join(d, name), "rb") as f:
        return f.read().decode("latin-1")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_refbatch_reproduces_the_reference_batch_sizes(case, name):
    d, ix, batch, sizes = case
    m, ram, frames = CONFIGS[name]
    rb = capi.RefBatcher(ix, 12, 7, frames, memory_gib=m, threads=1, ram=ram)
    bounds = rb.boundaries(batch, True)
    assert list(np.diff(bounds)) == sizes[name]


def _oracle_batched(ix, batch, bounds, frames, closed_form):
    rows, ca, cu, nq = [], None, None, 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        part = batch.slice(a, b)
        res, n = helpers.oracle_identify(ix, part, 12, 7, frames, closed_form=closed_form)
        for r in range(part.n):          # row by row: the dense matrix of a batch is ~1 GB here
            t = np.flatnonzero(res.M[r, 1:] > 0) + 1
            rows.append((t.astype(np.uint32), res.M[r, t].astype(np.float32)))
        ca = res.count_all if ca is None else ca + res.count_all
        cu = res.count_unique if cu is None else cu + res.count_unique
        nq += n
        del res
    return rows, ca, cu, nq


@pytest.mark.parametrize("name,closed_form", [("m1", False), ("m1", True), ("m2", True), ("m1_six", True)])
def test_oracle_batch_by_batch_equals_the_reference(case, name, closed_form):
    d, ix, batch, sizes = case
    m, ram, frames = CONFIGS[name]
    bounds = [0] + list(np.cumsum(sizes[name]))
    rows, ca, cu, nq = _oracle_batched(ix, batch, bounds, frames, closed_form)
    text, prof = helpers.render(ix, batch, rows, ca, cu, nq, "jsonl", 12, 7, frames, 0.0, 100)
    assert text == _text(d, "out_%s.jsonl" % name)
    assert prof == _text(d, "prof_%s.csv" % name)


def test_fixture_is_sensitive_to_the_boundaries(case):
    d, ix, batch, sizes = case
    a, b = _text(d, "out_m1.jsonl").split("\n"), _text(d, "out_m2.jsonl").split("\n")
    assert len(a) == len(b) and sum(x != y for x, y in zip(a, b)) >= 10   # same reads, other batches: other digits
    # ... and the oracle with the wrong boundaries does not reproduce the file
    rows, ca, cu, nq = _oracle_batched(ix, batch, [0, 3000, 6000], 3, True)
    text, _ = helpers.render(ix, batch, rows, ca, cu, nq, "jsonl", 12, 7, 3, 0.0, 100)
    assert text != _text(d, "out_m1.jsonl")


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONFIGS))
def test_python_host_byte_identical_across_batches(case, name):
    from kasa_amd import identify
    d, ix, batch, sizes = case
    m, ram, frames = CONFIGS[name]
    run = identify.Identify(ix, 0, 12, 7, frames, 0.0, 100, "jsonl")
    text, prof, _ = run.run(batch, True, memory_gib=m, threads=1, ram=ram)
    assert run.batch_sizes == sizes[name]
    assert text == _text(d, "out_%s.jsonl" % name)          # scores included, nothing stripped
    assert prof == _text(d, "prof_%s.csv" % name)
    run.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONFIGS))
def test_cpp_host_byte_identical_across_batches(case, name, tmp_path):
    from kasa_amd import build as hipbuild
    d, ix, batch, sizes = case
    m, ram, frames = CONFIGS[name]
    exe = hipbuild.build_host()
    out, prof = str(tmp_path / "out.jsonl"), str(tmp_path / "prof.csv")
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i",
           os.path.join(d, "reads.fastq.gz"), "-q", out, "-p", prof, "--jsonl", "-b", "100", "-m", str(m), "-n", "1", "-v"]
    if ram:
        cmd.append("-r")
    if frames == 6:
        cmd.append("--six")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = [int(x) for line in r.stdout.splitlines() if line.startswith("OUT: Batch of ") for x in [line.split()[3]]]
    assert got == sizes[name]
    assert _text(str(tmp_path), "out.jsonl") == _text(d, "out_%s.jsonl" % name)
    assert _text(str(tmp_path), "prof.csv") == _text(d, "prof_%s.csv" % name)
