"""What the return leg of the partitioned exchange carries: the bench's index (4.2e8 records) in 8 range partitions on ONE device
(partition.LocalExchange, device-resident), 2 M bench reads; bytes of records packed for the wire (kasa_batch_records_pack) against
whole records, per query, and the batch bit-equal to the unpartitioned run.
    python tools/wire_probe.py [--reads 2000000] [--parts 8]
Prints one JSON object (profiles/r06_wire_probe.json)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--parts", type=int, default=8)
    args = ap.parse_args()
    import numpy as np
    from kasa_amd import capi, partition, synth
    g = synth.genomes(1400, 300_000, seed=11)
    ix = synth.index_from_genomes(g, device=0, K=12)
    reads = synth.reads_from_genomes(g, args.reads, 150, seed=1000)
    dix = capi.DeviceIndex(ix, 0, check_trie=False)
    ctx = capi.Context(dix, 12, 7, 3)
    ctx.run_batch(reads.bases, reads.offsets, True)
    off, tax, sc = ctx.scores()
    limbs = ctx.profile_limbs().copy()
    nq = ctx.n_kmers
    ctx.close(); dix.close()
    parts, cuts = partition.split_index(ix, args.parts)
    res = {"reads": reads.n, "queries": int(nq), "index_records": int(ix.n), "partitions": args.parts}
    for packed in (True, False):
        ex = partition.LocalExchange(parts, cuts, 12, 7, 3, device_resident=True, packed=packed)
        ex.run_batch(reads)                                          # (sizes the buffers)
        ex.owner.profile_reset()
        t0 = time.perf_counter()
        c2 = ex.run_batch(reads)
        c2.synchronize()
        dt = time.perf_counter() - t0
        o2, t2, s2 = c2.scores()
        same = bool(np.array_equal(off, o2) and np.array_equal(tax, t2) and np.array_equal(sc.view(np.uint32), s2.view(np.uint32)) and np.array_equal(limbs, c2.profile_limbs()))
        res["packed" if packed else "whole"] = {"batch_s": round(dt, 4), "bit_equal_to_the_unpartitioned_run": same,
                                                "wire_bytes": int(ex.wire_bytes) if packed else int(nq) * 32, "bytes_per_query": (ex.wire_bytes / nq) if packed else 32.0}
        ex.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
