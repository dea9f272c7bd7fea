"""Do two batches in flight (two contexts, two streams, one index) finish sooner than one after the other?
The sort is bandwidth-bound, the group / score kernels are issue- and latency-bound: they might share the chip."""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from kasa_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
g = synth.genomes(1400, 300_000, seed=11)
ix = synth.index_from_genomes(g, device=0, K=12)
dix = capi.DeviceIndex(ix, 0, check_trie=False)
ctxs, batches = [], []
for i in range(2):
    rd = synth.reads_from_genomes(g, n, 150, seed=1000 + i)
    c = capi.Context(dix, 12, 7, 3)
    c.upload(rd.bases, rd.offsets)
    ctxs.append(c); batches.append(rd)

def step(c):
    c.encode(); c.sort_and_range(); c.lookup_score(True, False); c.synchronize()

for c in ctxs:
    step(c)                                   # warm-up: buffers sized
K = 3
t0 = time.perf_counter()
for _ in range(K):
    for c in ctxs:
        step(c)
seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=lambda c=c: [step(c) for _ in range(K)]) for c in ctxs]
for t in th: t.start()
for t in th: t.join()
par = time.perf_counter() - t0
print(f"{n} reads per batch, {2 * K} batches: one after the other {seq * 1e3 / (2 * K):.1f} ms per batch, two in flight {par * 1e3 / (2 * K):.1f} ms per batch")
