"""Shape of the event records at bench size (diagnostic; prints counters of kasa_debug_record_stats)."""
import ctypes as C
import json
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from kasa_amd import capi, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = synth.genomes(1400, 300_000, seed=11)
ix = synth.index_from_genomes(g, device=0, K=12)
reads = synth.reads_from_genomes(g, n_reads, 150, seed=1000)
dix = capi.DeviceIndex(ix, 0, check_trie=False)
ctx = capi.Context(dix, 12, 7, 3)
ctx.run_batch(reads.bases, reads.offsets)
out = np.zeros(32, np.uint64)
L = capi.lib()
L.kasa_debug_record_stats.argtypes = [C.c_void_p, C.c_void_p]
rc = L.kasa_debug_record_stats(ctx.h, out.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
names = ["live", "q_pool", "q_split", "q_sat", "segs", "other_segs", "other_pc1", "other_pc2", "other_pc3+", "-", "rec_split", "rec_sat",
         "sum_wave_max_nMore", "sum_wave_max_nseg", "q_monotone", "waves", "other_pc3+_plain", "q_norec", "records", "pool_segs",
         "q_other0", "q_other1", "q_other2", "q_other3+", "nseg1", "nseg2", "nseg3", "nseg4", "nseg5-8", "nseg9+", "sum_wave_max_other", "waves_with_split"]
d = {n: int(v) for n, v in zip(names, out)}
print(json.dumps(d, indent=1))
json.dump(d, open("gpurun_out/record_stats.json", "w"), indent=1)
