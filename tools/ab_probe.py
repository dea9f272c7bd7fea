"""A/B runs of one kernel choice against another on the bench workload (10 M x 150 bp reads, 5 GB index), one process, the
same resident batch: stage times per kasa_ctx_debug flag set, then kasa_batch_rank alone.  Usage:
    python tools/ab_probe.py [--reads N] [--flags 0,524288] [--rank-flags 0,1048576] [--rounds 2]
Prints one JSON object per measurement."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--taxa", type=int, default=1400)
    ap.add_argument("--genome-len", type=int, default=300_000)
    ap.add_argument("--flags", default="0,524288")
    ap.add_argument("--rank-flags", default="0,1048576")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    from kasa_amd import capi, report, synth
    g = synth.genomes(args.taxa, args.genome_len, seed=11)
    ix = synth.index_from_genomes(g, device=0, K=12)
    dix = capi.DeviceIndex(ix, 0, check_trie=False)
    ctx = capi.Context(dix, 12, 7, 3)
    reads = synth.reads_from_genomes(g, args.reads, 150, seed=1000)
    bases = capi.pinned_empty(reads.bases.shape[0], np.uint8)
    offsets = capi.pinned_empty(reads.offsets.shape[0], np.int64)
    bases[:] = reads.bases
    offsets[:] = reads.offsets

    def step():
        ctx.upload(bases, offsets)
        ctx.encode()
        ctx.sort_and_range()
        ctx.lookup_score(True, False)

    step(); step()                                                          # sizes every buffer
    ctx.profile_reset()
    step()
    ref = ctx.profile_limbs().copy()
    for rnd in range(args.rounds):
        for fl in [int(x) for x in args.flags.split(",")]:
            ctx.debug_flags(fl)
            ctx.profile_reset()
            step()
            same = bool((ctx.profile_limbs() == ref).all()) if rnd == 0 else None
            ctx.stage_reset()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            st = {k: round(v[0] / max(1, v[1]), 3) for k, v in ctx.stage_ms().items()}
            km = {k: round(v[0] / max(1, v[1]), 3) for k, v in ctx.kernel_ms().items() if v[1]}
            print(json.dumps({"flags": fl, "ms_per_step_wall": round(dt * 1e3, 2), "stage_ms": st, "kernel_ms": km, "group_tiles": ctx.group_tiles(),
                              "profile_same_as_flags0": same}), flush=True)
    ctx.debug_flags(0)
    if not args.rank_flags:
        ctx.close(); dix.close()
        return
    step()
    den, rclass = report.rank_denominators(ix.freq_at(12), reads.lengths, ix.K, False)
    den = np.ascontiguousarray(den, dtype=np.float64)
    rclass = np.ascontiguousarray(rclass, dtype=np.uint32)
    L = capi.lib()
    first = None
    for rnd in range(args.rounds):
        for fl in [int(x) for x in args.rank_flags.split(",")]:
            ctx.debug_flags(fl)
            n_ent, n_flag = C.c_uint64(0), C.c_uint32(0)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                capi._check(L.kasa_batch_rank(ctx.h, capi._p(den), C.c_uint32(den.shape[0]), capi._p(rclass), C.c_float(0.0), C.c_uint32(3),
                                              C.byref(n_ent), C.byref(n_flag)))
                ts.append((time.perf_counter() - t0) * 1e3)
            meta, ent, _ = ctx.rank(den, rclass, 0.0, 3)
            # entries in read order (slab positions depend on the scheduling)
            order = meta[:, 0].astype(np.int64)
            cnt = (meta[:, 1] & 0x7FFFFFFF).astype(np.int64)
            pick = np.repeat(order, cnt) + (np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt))
            flat = ent[pick].tobytes()
            if first is None:
                first = (flat, cnt.copy())
            print(json.dumps({"rank_flags": fl, "rank_ms": [round(t, 2) for t in ts], "entries": int(n_ent.value), "left_to_host": int(n_flag.value),
                              "same_as_first": bool(flat == first[0] and (cnt == first[1]).all())}), flush=True)
    ctx.close()
    dix.close()


if __name__ == "__main__":
    main()
