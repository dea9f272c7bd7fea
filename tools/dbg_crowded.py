import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kasa_amd import capi, synth
from oracle import oracle
n_taxa = int(os.environ.get("T", "200")); n_reads = int(os.environ.get("R", "20000"))
g = synth.genomes_crowded(n_taxa, 300_000, seed=11)
ix = synth.index_from_genomes(g, K=12)
batch = synth.reads_from_genomes(g, int(os.environ.get("DRAW", str(n_reads))), 150, seed=1000).slice(0, n_reads)
res, nq = oracle.identify_batch(ix, batch.bases, batch.offsets, oracle.params(12, 7, 3, K=12), True)
rows, cols = np.nonzero(res.M[:, 1:] > 0)
ref = res.M[rows, cols + 1].astype(np.float32).view(np.uint32)
dix = capi.DeviceIndex(ix)
for flags in (0, 1, 1 | 8192):
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.debug_flags(flags)
    ctx.run_batch(batch.bases, batch.offsets, True)
    o, t, v = ctx.scores()
    same_struct = np.array_equal(t, (cols + 1).astype(np.uint32))
    bad = np.flatnonzero(v.view(np.uint32) != ref) if same_struct else None
    st = ctx.batch_stats()
    if bad is not None and len(bad):
        rd = np.searchsorted(o, bad, side="right") - 1
        urd = np.unique(rd)
        print(flags, "struct", same_struct, "bad cells", len(bad), "bad reads", len(urd), "general", st["general_reads"], "first", urd[:5], flush=True)
        r = int(urd[0]); lo, hi = int(o[r]), int(o[r + 1])
        print("   read", r, "cells", hi - lo, "diff", [(int(t[i]), float(v[i]), float(res.M[r, t[i]])) for i in bad[rd == r][:4]])
        ulp = np.abs(v.view(np.int32)[bad].astype(np.int64) - ref.view(np.int32)[bad].astype(np.int64))
        print("   ulp diffs: max", int(ulp.max()), "hist", np.bincount(np.minimum(ulp, 5))[:6], "cells per bad read", np.bincount(np.bincount(rd - rd.min()))[:6])
    else:
        print(flags, "struct", same_struct, "all bits equal" if bad is not None else "", "general", st["general_reads"], flush=True)
    ctx.close()
