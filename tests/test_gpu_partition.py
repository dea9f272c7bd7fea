"""Range-partitioned index (C5): the batch run slice by slice against partitions of the index must equal the run
against the whole index bit for bit -- per-read rows, profile tables, and the reference's own files."""
import os

import numpy as np
import pytest

from kasa_amd import capi, partition, reads, report
from tests import helpers
from tests.test_gpu_parity import assert_csr_equal, csr_rows, synthetic_world

pytestmark = pytest.mark.gpu


def _whole(ix, batch, kh, kl, frames, unique=False):
    dix = capi.DeviceIndex(ix)
    ctx = capi.Context(dix, kh, kl, frames)
    ctx.run_batch(batch.bases, batch.offsets, True, unique=unique, seg_read=batch.seg_read, n_reads=batch.n)
    out = (ctx.scores(), ctx.profile_limbs().copy())
    ctx.close(); dix.close()
    return out


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
@pytest.mark.parametrize("n_parts", [2, 3, 7])
def test_golden_index_in_partitions(n_parts, resident):
    assert capi.device_count() > 0
    d, ix = helpers.load_case("pairs")
    batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
    parts, cuts = partition.split_index(ix, n_parts)
    assert sum(p.n for p in parts) == ix.n and all(p.n > 0 for p in parts)
    (off, tax, sc), limbs = _whole(ix, batch, 12, 7, 3)
    ex = partition.LocalExchange(parts, cuts, 12, 7, 3, device_resident=resident)
    ctx = ex.run_batch(batch)
    o2, t2, s2 = ctx.scores()
    assert np.array_equal(off, o2) and np.array_equal(tax, t2) and np.array_equal(sc.view(np.uint32), s2.view(np.uint32))
    assert np.array_equal(limbs, ctx.profile_limbs())
    # and through the writers: the reference's own file
    w = report.ReadWriter("jsonl", ix.content.names, ix.content.taxids, 100)
    text = w.header()
    for r in range(batch.n):
        lo, hi = int(o2[r]), int(o2[r + 1])
        rk = report.rank_read(t2[lo:hi], s2[lo:hi], int(batch.lengths[r]), ix.freq_at(12), 12, 7, 3, 0.0, 100)
        text += w.read(r, batch.names[r], int(batch.lengths[r]), rk)
    text += w.footer()
    with open(os.path.join(d, "out_b100.jsonl"), "rb") as f:
        assert text == f.read().decode("latin-1")
    ex.close()


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
@pytest.mark.parametrize("case", [(12, 7, 3, False, 12), (12, 7, 6, True, 12), (25, 7, 3, False, 25), (10, 5, 3, False, 12)])
def test_synthetic_partitions(case, resident):
    kh, kl, frames, unique, K = case
    ix, batch = synthetic_world(83, 10, 9000, 2500, K=K)
    parts, cuts = partition.split_index(ix, 4)
    (off, tax, sc), limbs = _whole(ix, batch, kh, kl, frames, unique)
    ex = partition.LocalExchange(parts, cuts, kh, kl, frames, device_resident=resident)
    ctx = ex.run_batch(batch, unique=unique)
    assert_csr_equal(csr_rows(*ctx.scores()), csr_rows(off, tax, sc))
    assert np.array_equal(limbs, ctx.profile_limbs())
    ex.close()


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from kasa_amd import capi, partition, dist as kdist
from tests.test_gpu_parity import synthetic_world
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ix, batch = synthetic_world(97, 8, 7000, 1800)
parts, cuts = partition.split_index(ix, world)
a, b = kdist.shard_bounds(batch.n, rank, world)
mine = batch.slice(a, b)
dix = capi.DeviceIndex(parts[rank])                       # this rank holds ONE partition
owner = capi.Context(dix, 12, 7, 3)
worker = partition.Worker(dix, 12, 7, 3)
# rank 0 runs its reads as one batch, rank 1 cuts its own into two: the ranks hold different numbers of batches
cut = [0, mine.n] if rank == 0 else [0, mine.n // 3, mine.n]
offs, taxs, scs, base = [np.zeros(1, dtype=np.uint64)], [], [], 0
for ctx in kdist.partitioned_batches(owner, worker, cuts, ix.K, [mine.slice(x, y) for x, y in zip(cut[:-1], cut[1:])]):
    o, t, v = ctx.scores()
    offs.append(o[1:] + np.uint64(base)); taxs.append(t); scs.append(v)
    base += int(o[-1])
off, tax, sc = np.concatenate(offs), np.concatenate(taxs), np.concatenate(scs)
limbs = kdist.allreduce_limbs(owner.profile_limbs())
np.savez(os.path.join(sys.argv[2], f"rank{rank}.npz"), off=off, tax=tax, sc=sc, limbs=limbs, a=a, b=b, cut=np.asarray(cut))
dist.barrier()
dist.destroy_process_group()
"""


RCCL_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from kasa_amd import capi, partition, dist as kdist
from tests.test_gpu_parity import synthetic_world
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ix, batch = synthetic_world(97, 8, 7000, 1800)
parts, cuts = partition.split_index(ix, 1)
dix = capi.DeviceIndex(parts[0])
owner = capi.Context(dix, 12, 7, 3)
worker = partition.Worker(dix, 12, 7, 3)
owner.queries = owner.records = None                      # the RCCL path must never stage through the host
worker.ctx.queries = worker.ctx.records = None
ctx = kdist.partitioned_batch(owner, worker, cuts, ix.K, batch)
off, tax, sc = ctx.scores()
np.savez(os.path.join(sys.argv[2], "rccl.npz"), off=off, tax=tax, sc=sc, limbs=ctx.profile_limbs())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("case", [(12, 7, 12), (25, 7, 25), (12, 9, 25)], ids=["narrow_records", "wide_records", "narrow_records_of_a_128_bit_index"])
def test_records_on_the_wire(case):
    """The return leg of the exchange: kasa_batch_records_pack writes exactly the bytes partition.pack_records (the wire format's
    statement in numpy) makes of the same records -- a byte of classes per four queries, then the words a record really uses --
    kasa_batch_records_unpack gives the records back with their unused words zero, and a batch whose records crossed that way
    (LocalExchange, the default) scores bit for bit like one whose whole records did.  SURVEY 8(e): 12-20 bytes per query."""
    kh, kl, K = case
    ix, batch = synthetic_world(61, 10, 4000, 600, K=K)
    parts, cuts = partition.split_index(ix, 3)
    dix = capi.DeviceIndex(parts[1])
    owner = capi.Context(dix, kh, kl, 3)
    owner.upload(batch.bases, batch.offsets)
    owner.encode()
    owner.sort_and_range()
    ptr, n, kb = owner.queries_device()
    starts = owner.slice_starts(cuts)
    w = partition.Worker(dix, kh, kl, 3)
    nq = int(starts[2] - starts[1])
    rp, nrw, pp, npw = w.group_slice_device(ptr + int(starts[1]) * kb, nq)
    rw = owner.rec_words
    assert nrw == nq * rw and nq > 1000
    rec, _ = w.ctx.records()
    rec = np.asarray(rec).reshape(-1, rw)
    want = partition.pack_records(rec, rw)
    nb = w.ctx.records_pack_size(rp, nq)
    assert nb == want.nbytes
    buf = capi.DeviceBuffer(nb)
    w.ctx.records_pack(rp, nq, buf.ptr, nb)
    assert np.array_equal(buf.read(), want)
    back = capi.DeviceBuffer(nq * rw * 4)
    owner.records_unpack(buf.ptr, nb, nq, back.ptr)
    assert np.array_equal(back.read().view(np.uint32).reshape(-1, rw), partition.unpack_records(want, nq, rw))
    matched = (rec[:, 2] & 31) != 0
    assert nb < 0.8 * nq * rw * 4 and 0.5 < matched.mean() < 1.0, (nb, nq * rw * 4, matched.mean())
    with pytest.raises(RuntimeError):
        owner.records_unpack(buf.ptr, nb - 4, nq, back.ptr)               # (the classes announce more words than the buffer holds)
    buf.close(); back.close(); w.close(); owner.close(); dix.close()
    (off, tax, sc), limbs = _whole(ix, batch, kh, kl, 3)
    for resident in (False, True):
        for packed in (True, False):
            ex = partition.LocalExchange(parts, cuts, kh, kl, 3, device_resident=resident, packed=packed)
            ctx = ex.run_batch(batch)
            o2, t2, s2 = ctx.scores()
            assert np.array_equal(off, o2) and np.array_equal(tax, t2) and np.array_equal(sc.view(np.uint32), s2.view(np.uint32))
            assert np.array_equal(limbs, ctx.profile_limbs())
            assert (ex.wire_bytes < 0.8 * ex.whole_bytes) if packed else ex.wire_bytes == 0
            ex.close()


def test_device_resident_exchange_over_rccl(tmp_path):
    """kasa_amd/dist.py:partitioned_batch on the `nccl` backend (RCCL): the slices and the records travel as device
    tensors through all_to_all_single and the kasa_batch_*_device entry points -- Context.queries()/records() are
    removed in the worker process, so any host staging would fail.  One rank (this box has one GPU; RCCL refuses two
    ranks on one device): the collective degenerates, the plumbing does not."""
    import socket
    import subprocess
    import sys
    assert capi.device_count() > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, str(script), root, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(tmp_path / "rccl.npz")
    ix, batch = synthetic_world(97, 8, 7000, 1800)
    (off, tax, sc), limbs = _whole(ix, batch, 12, 7, 3)
    assert np.array_equal(off, z["off"]) and np.array_equal(tax, z["tax"]) and np.array_equal(sc.view(np.uint32), z["sc"].view(np.uint32))
    assert np.array_equal(limbs, z["limbs"])


def test_two_ranks_one_partition_each(tmp_path):
    """The real exchange code (kasa_amd/dist.py:partitioned_batch) with two processes, each holding half of the index
    and half of the reads; gloo carries the slices (both processes use this box's one GPU)."""
    import socket
    import subprocess
    import sys
    assert capi.device_count() > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), root, str(tmp_path)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    ix, batch = synthetic_world(97, 8, 7000, 1800)
    dix = capi.DeviceIndex(ix)
    total = None
    for rank in range(2):
        z = np.load(tmp_path / f"rank{rank}.npz")
        part = batch.slice(int(z["a"]), int(z["b"]))
        ctx = capi.Context(dix, 12, 7, 3)
        cut = [int(x) for x in z["cut"]]
        offs, taxs, scs, base = [np.zeros(1, dtype=np.uint64)], [], [], 0
        for x, y in zip(cut[:-1], cut[1:]):                       # the same batches on the whole index
            sub = part.slice(x, y)
            ctx.run_batch(sub.bases, sub.offsets, True)
            o, t, v = ctx.scores()
            offs.append(o[1:] + np.uint64(base)); taxs.append(t); scs.append(v)
            base += int(o[-1])
        off, tax, sc = np.concatenate(offs), np.concatenate(taxs), np.concatenate(scs)
        assert np.array_equal(off, z["off"]) and np.array_equal(tax, z["tax"])
        assert np.array_equal(sc.view(np.uint32), z["sc"].view(np.uint32))
        from kasa_amd import dist as kdist
        ca, cu, _ = kdist.limbs_to_tables(ctx.profile_limbs(), 6, ix.content.n_taxa)
        total = (ca, cu) if total is None else (total[0] + ca, total[1] + cu)
        ca_r, cu_r, _ = kdist.limbs_to_tables(z["limbs"], 6, ix.content.n_taxa)
        ctx.close()
    assert np.array_equal(cu_r, total[1])
    np.testing.assert_allclose(ca_r, total[0], rtol=1e-12)
    dix.close()


def _bench(args, share=True, timeout=900):
    import json
    import subprocess
    import sys
    assert capi.device_count() > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if share:
        env["KASA_BENCH_SHARE_GPU"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (fresh processes, before
    anything touches a GPU) and rank 0 prints the one JSON line.  On a one-GPU box both ranks share device 0 and the
    reduce runs over gloo (KASA_BENCH_SHARE_GPU=1: the multi-rank code path, not a measurement).  The workload is
    BASELINE.json configs[3] in small: a fixed total of reads, every rank's share resident in HBM and taken in batches."""
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "10000", "--total-reads", "50000",
                  "--taxa", "8", "--genome-len", "20000", "--no-cpu"])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["roofline"]["kernel"]
    assert out["scaling"] == "strong" and out["config"]["batches_per_step"] == 3 and out["config"]["reads_per_gpu"] == 25000
    assert out["config"]["total_reads"] == 50000 and 0.5 < out["identified_fraction"] <= 1.0


def test_bench_same_step_at_every_n(tmp_path):
    """The default multi-GPU line: every rank runs the SAME resident batch loop as N = 1 (weak: --reads per rank, the reduce
    is the only difference) and the line carries `c4` (BASELINE.json configs[3] in small: a fixed total, in batches), the
    per-rank step times and the reduce time.  N = 1 with --total-reads goes through the same loop in several batches."""
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "8000", "--c4-reads", "30000",
                  "--taxa", "8", "--genome-len", "20000", "--no-cpu"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["batches_per_step"] == 1 and out["config"]["reads_per_gpu"] == 8000
    assert out["config"]["total_reads"] == 16000 and len(out["rank_step_ms"]["per_rank"]) == 2 and out["reduce_ms_per_step"] > 0
    assert out["upload_ms_per_batch"] > 0 and 0.5 < out["identified_fraction"] <= 1.0
    c4 = out["c4"]
    assert c4["scaling"] == "strong" and c4["config"]["total_reads"] == 30000 and c4["config"]["batches_per_step"] == 2 and c4["value"] > 0
    c2s = out["c2_strong"]                                        # BASELINE.json's metric read literally: --reads reads in all over the ranks
    assert c2s["scaling"] == "strong" and c2s["config"]["total_reads"] == 8000 and c2s["config"]["reads_per_gpu"] == 4000 and c2s["value"] > 0
    assert out["config"]["rccl_ranks_tested"] == 1 and out["attempts"] == 1 and out["retried"] is False
    assert out["runtime"]["hip_built"] and out["runtime"]["hip_runtime"]
    one = _bench(["--steps", "2", "--warmup", "1", "--reads", "8000", "--total-reads", "20000", "--taxa", "8", "--genome-len", "20000",
                  "--no-cpu", "--no-e2e", "--no-secondary", "--no-tertiary", "--no-quaternary", "--no-pmc"], share=False)
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["config"]["batches_per_step"] == 3 and one["reduce_ms_per_step"] == 0
    assert 0.5 < one["identified_fraction"] <= 1.0


def test_bench_default_line_carries_every_leg(tmp_path):
    """N = 1 in small: the headline (configs[1]), `secondary` (configs[2], 128-bit index), `tertiary` (the crowded index), `quaternary` (long reads), the
    PCIe-inclusive and the file-to-file rates, the CPU baseline with its one-thread rate, the dominant kernel's HBM traffic from
    counter passes run as children -- all in the one JSON line."""
    out = _bench(["--steps", "1", "--warmup", "1", "--reads", "30000", "--taxa", "8", "--genome-len", "20000",
                  "--cpu-sample", "5000", "--cpu-sample-parallel", "20000", "--f2f-settle", "0", "--crowded-reads", "10000", "--long-reads-n", "300"], share=False)
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and out["dtype"] == "u64"
    assert out["secondary"]["dtype"] == "u128" and out["secondary"]["value"] > 0
    assert out["tertiary"]["value"] > 0 and out["tertiary"]["config"]["database"] == "crowded" and out["tertiary"]["config"]["reads_per_gpu"] == 10000
    assert out["roofline"]["traffic_source"] and out["roofline"]["second_bound"]["bound"] == "scatter"
    assert out["roofline"]["traffic"] is None or out["roofline"]["traffic"] > 0
    e = out["e2e"]
    assert e["pcie_inclusive_reads_per_s"] > 0 and e["file_to_file_reads_per_s"] > 0 and e["batches"] >= 1
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["single_thread_value"] > 0 and c["speedup_over_1"] > 0
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    assert out["attempts"] == 1 and out["retried"] is False and out["runtime"]["hip_runtime"]       # (no silent second attempt)
    assert "score_dense_kernel" in out["tertiary"]["kernels"] or out["tertiary"]["batch"]["dense_reads"] == 0
    assert out["tertiary"]["roofline"]["traffic_source"]
    q = out["quaternary"]                                            # long reads: 10 kb reads; one contig (replayed from sorted events)
    assert q["reads_10kb"]["reads"] == 300 and q["reads_10kb"]["kmers_per_s"] > 0
    assert q["contig_different_genomes"]["replay_reads"] == 1 and q["contig_one_genome_32_times"]["replay_events"] > 0
    if out["roofline"].get("third_bound"):
        assert out["roofline"]["third_bound"]["bound"] == "issue" and out["roofline"]["third_bound"]["predicted_ms"] > 0


def test_c_abi_reduce_with_a_communicator_of_its_own(tmp_path):
    """kasa_amd/dist.py:rccl_communicator (ncclUniqueId over torch.distributed, ncclCommInitRank by ctypes) + kasa_profile_allreduce
    -- what bench.py's multi-GPU step uses -- with the one rank this box has: the tables come back unchanged."""
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from kasa_amd import capi, formats, reads, dist as kdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
d = os.path.join(sys.argv[1], "tests", "golden", "pairs")
ix = formats.load_index(os.path.join(d, "idx"), os.path.join(d, "content.txt"))
batch = reads.parse_reads(os.path.join(d, "reads.fastq"))
dix = capi.DeviceIndex(ix); ctx = capi.Context(dix, 12, 7, 3)
ctx.run_batch(batch.bases, batch.offsets, True)
before = ctx.profile_limbs().copy()
comm, n = kdist.rccl_communicator(0, 1)
assert n == 1 and comm
ctx.profile_allreduce(comm)
ctx.synchronize()
ca0, cu0, _ = ctx.profile()
assert np.array_equal(cu0, before.reshape(-1, 6)[:, 0].reshape(cu0.shape))
ctx.profile_allreduce(comm)                                   # idempotent with one rank
ca1, cu1, _ = ctx.profile()
assert np.array_equal(cu0, cu1) and np.array_equal(ca0, ca1) and cu0.sum() > 0
kdist.rccl_destroy(comm)
dist.destroy_process_group()
print("ok")
""")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script), root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("gpus,share", [(2, True), (1, False)], ids=["two_ranks_share_the_gpu_gloo", "one_rank_rccl_device_resident"])
def test_bench_partitioned(gpus, share):
    """bench.py --partitioned (BASELINE.json configs[4] in small): every rank synthesises its slice of the index on the device
    (the genomes' records of its prefix range + random filler), sizes its batch from the free HBM and runs the exchange."""
    out = _bench(["--partitioned", "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--reads", "4000", "--part-records", "3e6",
                  "--taxa", "8", "--genome-len", "20000"], share=share)
    assert out["n_gpus"] == gpus and out["value"] > 0 and out["config"]["slice_records"] > 1_000_000
    assert 0.3 < out["identified_fraction"] <= 1.0
    x = out["exchange_bytes_per_step_this_rank"]
    if gpus > 1:
        assert x["queries_sent"] > 0 and x["records_received"] > 0


def test_index_beyond_2_32_records_in_device_partitions():
    """Positions in an index are 32-bit (kasa_index_create refuses more than 2^32 - 1 records): ONE device holds a larger index
    as several range partitions, every one an index of its own, behind partition.LocalExchange (device-resident slices and
    records).  Here: 4.5e9 records (54 GB as a file) -- the records of 40 genomes plus random filler, synthesised on the
    device per quarter of the prefix range -- as four partitions and as two: per-read scores and profile must not depend on
    the cut (bit for bit), every read must have found its genome's records (hits at k = 12 can only come from them), and
    the batch must have met filler (hits at k = 7 from taxa the read does not come from)."""
    assert capi.device_count() > 0
    import torch
    from kasa_amd import synth
    free, _ = capi.device_memory(0)
    if free < 200e9:
        pytest.skip("needs 200 GB of free HBM")
    dev = torch.device("cuda", 0)
    n_taxa, per_quarter = 40, 1_120_000_000
    g = synth.genomes(n_taxa, 60_000, seed=5)
    ix = synth.index_from_genomes(g, device=0)
    parts4, cuts4 = partition.split_index(ix, 4)
    parts2, cuts2 = partition.split_index(ix, 2)
    assert int(cuts2[1]) == int(cuts4[2])                          # the halves are pairs of quarters
    quarters = []
    for q in range(4):
        lo = int(cuts4[q]) << 30
        hi = (int(cuts4[q + 1]) if q < 3 else (1 << 30)) << 30
        mine = parts4[q]
        gen = torch.Generator(device=dev)
        gen.manual_seed(777 + q)
        km = torch.empty(mine.n + per_quarter, dtype=torch.int64, device=dev)
        td = torch.empty(mine.n + per_quarter, dtype=torch.int32, device=dev)
        km[:mine.n] = torch.from_numpy(mine.kmer.astype(np.int64)).to(dev)
        td[:mine.n] = torch.from_numpy(mine.taxid.astype(np.int32)).to(dev)
        for a in range(mine.n, mine.n + per_quarter, 1 << 28):
            b = min(mine.n + per_quarter, a + (1 << 28))
            km[a:b] = torch.randint(lo, hi, (b - a,), dtype=torch.int64, device=dev, generator=gen)
            td[a:b] = torch.randint(100, 100 + n_taxa, (b - a,), dtype=torch.int32, device=dev, generator=gen)
        td, order = torch.sort(td, stable=True)                    # (kmer, taxid) order
        km = km[order]
        km, order = torch.sort(km, stable=True)
        td = td[order]
        del order
        keep = torch.ones(km.shape[0], dtype=torch.bool, device=dev)
        keep[1:] = (km[1:] != km[:-1]) | (td[1:] != td[:-1])
        km, td = km[keep], td[keep]
        del keep
        rec = torch.empty((int(km.shape[0]), 12), dtype=torch.uint8, device=dev)   # the index file's records: {u64 kmer, u32 taxid}
        rec[:, :8] = km.view(torch.uint8).view(-1, 8)
        rec[:, 8:] = td.view(torch.uint8).view(-1, 4)
        del km, td
        quarters.append(rec)
    torch.cuda.synchronize()
    total = sum(int(r.shape[0]) for r in quarters)
    assert total > (1 << 32)
    batch = synth.reads_from_genomes(g, 20_000, 150, seed=9)

    def run(recs, cuts):
        dix = [capi.DeviceIndex.from_device_records(r.data_ptr(), int(r.shape[0]), 12, ix.content.taxids, 0) for r in recs]
        ex = partition.LocalExchange(dix, cuts, 12, 7, 3, device_resident=True, K=12)
        ctx = ex.run_batch(batch)
        out = (ctx.scores(), ctx.profile_limbs().copy(), ctx.profile())
        ex.close()
        return out

    (o4, t4, s4), limbs4, (ca, cu, _) = run(quarters, cuts4)
    halves = [torch.cat(quarters[0:2]), torch.cat(quarters[2:4])]
    del quarters
    torch.cuda.empty_cache()
    (o2, t2, s2), limbs2, _ = run(halves, cuts2)
    del halves
    torch.cuda.empty_cache()
    assert np.array_equal(o4, o2) and np.array_equal(t4, t2) and np.array_equal(s4.view(np.uint32), s2.view(np.uint32))
    assert np.array_equal(limbs4, limbs2)
    n_kmers = 3 * 20_000 * (150 // 3 - 12 + 1) - 2 * 20_000       # (frames 1 and 2 hold one window less: 39 + 38 + 38 per read)
    assert 0.60 * n_kmers < float(ca[0].sum()) <= n_kmers           # k = 12: the genomes' own records (0.99^36 = 70 % of the windows carry no substitution)
    assert float(ca[-1].sum()) > float(ca[0].sum())                 # k = 7: filler joins in
    per_read = np.diff(o4.astype(np.int64))
    assert per_read.min() >= 1 and per_read.mean() > 10            # rows hold the genome and filler taxa
