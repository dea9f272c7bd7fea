"""world_size-2 gloo test of the multi-GPU exchange: per-rank profile limbs summed with one all-reduce,
carries normalised afterwards; read sharding is contiguous and covers every read once."""
import os
import socket
import subprocess
import sys

import numpy as np

from kasa_amd import dist as kdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from kasa_amd import dist as kdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(100 + rank)
n_k, n_taxa = 3, 7
limbs = np.zeros((n_k * n_taxa, 6), dtype=np.uint64)
limbs[:, 0] = rng.integers(0, 1 << 40, size=n_k * n_taxa)
limbs[:, 1] = rng.integers(0, 1 << 20, size=n_k * n_taxa)
limbs[:, 2:] = rng.integers(0, 1 << 32, size=(n_k * n_taxa, 4))      # full 32-bit limbs: carries happen
out = kdist.allreduce_limbs(limbs)
np.save(os.path.join(sys.argv[2], f"in{rank}.npy"), limbs)
if rank == 0:
    np.save(os.path.join(sys.argv[2], "out.npy"), out)
dist.barrier()
dist.destroy_process_group()
"""


def test_shard_bounds_cover_all_reads():
    for n in (0, 1, 7, 10_000_001):
        for world in (1, 2, 3, 8):
            edges = [kdist.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1


def test_allreduce_limbs_gloo_world2(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=env))
    for p in procs:
        assert p.wait(timeout=180) == 0
    a, b = np.load(tmp_path / "in0.npy"), np.load(tmp_path / "in1.npy")
    out = np.load(tmp_path / "out.npy")
    assert np.array_equal(out, a + b)
    all_, uniq, tot = kdist.limbs_to_tables(out, 3, 7)
    assert np.array_equal(uniq.reshape(-1), a[:, 0] + b[:, 0])
    for i in range(21):
        va = sum(int(a[i, 2 + j]) << (32 * j) for j in range(4))
        vb = sum(int(b[i, 2 + j]) << (32 * j) for j in range(4))
        want = (va + vb) / 2.0 ** 64
        assert abs(all_.reshape(-1)[i] - want) <= 1e-12 * max(1.0, want)
