import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from kasa_amd import synth
from oracle import oracle
os.environ["KO_TIMING"] = "1"
g = synth.genomes(int(sys.argv[1]) if len(sys.argv) > 1 else 1400, 300_000, seed=11)
ix = synth.index_from_genomes(g)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
reads = synth.reads_from_genomes(g, n, 150, seed=1000)
iv = oracle.IndexView(ix)
p = oracle.params(12, 7, 3)
for T in (os.cpu_count(), 64, 16):
    t0 = time.perf_counter()
    oracle.identify_threaded(iv, reads.bases, reads.offsets, p, T)
    print("threads", T, "total", time.perf_counter() - t0, flush=True)
