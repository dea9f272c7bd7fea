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


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (fresh processes, before
    anything touches a GPU) and rank 0 prints the one JSON line.  On a one-GPU box both ranks share device 0 and the
    reduce runs over gloo (KASA_BENCH_SHARE_GPU=1: the multi-rank code path, not a measurement)."""
    import json
    import subprocess
    import sys
    assert capi.device_count() > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KASA_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "20000",
                        "--taxa", "8", "--genome-len", "20000", "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["roofline"]["kernel"]
