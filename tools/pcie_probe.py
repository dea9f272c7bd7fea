import sys, time, numpy as np
sys.path.insert(0, '.')
from kasa_amd import capi, synth
g = synth.genomes(1400, 300_000, seed=11)
ix = synth.index_from_genomes(g)
reads = synth.reads_from_genomes(g, 10_000_000, 150, seed=1000)
dix = capi.DeviceIndex(ix)
ctx = capi.Context(dix, 12, 7, 3)
for rep in range(2):
    t0 = time.perf_counter(); ctx.upload(reads.bases, reads.offsets); ctx.synchronize(); t1 = time.perf_counter()
    ctx.encode(); ctx.sort_and_range(); ctx.lookup_score(True); ctx.synchronize(); t2 = time.perf_counter()
    off, tax, sc = ctx.scores(); t3 = time.perf_counter()
    print("upload %.3f s (%.1f GB), device %.3f s, CSR download %.3f s (%.1f GB), total %.3f s -> %.2f M reads/s" % (
        t1 - t0, (reads.bases.nbytes + reads.offsets.nbytes) / 1e9, t2 - t1, t3 - t2, (off.nbytes + tax.nbytes + sc.nbytes) / 1e9,
        t3 - t0, 10 / (t3 - t0)))
