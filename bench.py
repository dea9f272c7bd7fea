#!/usr/bin/env python3
"""bench.py -- reads/s of the `identify` hot path on MI355X (BASELINE.json metric).

One step = one pass of the whole hot path (encode -> sort -> lookup -> group -> regroup -> score, per-read
CSR included) over one batch of synthetic reads that is already resident in HBM.  Workload at N=1:
BASELINE.json configs[1]: 10 M synthetic 150 bp reads against a ~5 GB k<=12 64-bit index (1400 taxa x
300 kb, sibling genomes 3 % apart; 1 % read errors; -k 12 7, three frames).  With --gpus N every rank holds
the whole index and its own 10 M reads (weak scaling, BASELINE.json configs[3]); the per-rank profile
tables are summed with one RCCL all-reduce per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--taxa G] [--genome-len L]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(ix, sample, k_high, k_low):
    """The CPU oracle (a port of the reference algorithm, single thread) on a bounded sample."""
    from oracle import oracle
    p = oracle.params(k_high, k_low, 3, K=ix.K)
    iv = oracle.IndexView(ix)
    t0 = time.perf_counter()
    km, rd = oracle.encode(sample.bases, sample.offsets, p)
    km, rd = oracle.sort_queries(km, rd)
    rs, rl = oracle.ranges(iv, p, km)
    oracle.compare(iv, p, km, rd, rs, rl, sample.n, True)
    dt = time.perf_counter() - t0
    return sample.n / dt, dt


def stage_gbps(stages, steps, n_q, n_bases, n_idx, rec_bytes, n_k):
    key = rec_bytes - 4
    algo = {"encode": n_bases + n_q * rec_bytes, "sort": 2 * n_q * rec_bytes, "lookup": n_q * rec_bytes + n_idx * rec_bytes,
            "group": n_q * (key + 1 + 4) + n_q * n_k * 8, "regroup": n_q * 8}
    out = {}
    for k, b in algo.items():
        ms = stages.get(k, (0.0,))[0] / max(1, steps)
        out[k] = (b / (ms * 1e-3) / 1e9) if ms > 0 else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--taxa", type=int, default=1400)
    ap.add_argument("--genome-len", type=int, default=300_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=600_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--profile-only", action="store_true", help="no per-read scores (kASA without -q)")
    ap.add_argument("--wide", action="store_true",
                    help="secondary measurement (BASELINE.json configs[2]): 128-bit index, -k 25 7; not the headline line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (one-GPU boxes): KASA_BENCH_SHARE_GPU=1 runs every rank on device 0 and carries the reduce over gloo,
    # because RCCL refuses two ranks on one device; it exercises the multi-rank code path, it is not a measurement
    share = os.environ.get("KASA_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from kasa_amd import capi, synth
    from kasa_amd import dist as kdist
    assert capi.device_count() > local_rank, "no HIP device for this rank"

    k_high, k_low = (25, 7) if args.wide else (12, 7)
    rec_bytes = 20 if args.wide else 12
    t0 = time.perf_counter()
    g = synth.genomes(args.taxa, args.genome_len, seed=11)
    ix = synth.index_from_genomes(g, device=local_rank, K=25 if args.wide else 12)
    log(f"[rank {rank}] index: {ix.n} records ({ix.n * rec_bytes / 1e9:.2f} GB on disk layout), "
        f"{ix.trie_prefix.shape[0]} prefixes, {time.perf_counter() - t0:.1f} s")
    t0 = time.perf_counter()
    reads = synth.reads_from_genomes(g, args.reads, args.read_len, seed=1000 + rank)
    log(f"[rank {rank}] reads: {reads.n} x {args.read_len} bp, {time.perf_counter() - t0:.1f} s")

    dix = capi.DeviceIndex(ix, local_rank, check_trie=True)
    ctx = capi.Context(dix, k_high, k_low, 3)
    ctx.upload(reads.bases, reads.offsets)         # inputs resident in HBM before the timed region
    want = not args.profile_only

    def step():
        ctx.encode()
        ctx.sort_and_range()
        ctx.lookup_score(want, False)
        if dist is not None:
            kdist.allreduce_limbs(ctx.profile_limbs(), device=None if share else "cuda")   # one RCCL sum of integer limbs (exact)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.profile_reset()
        step()
    fence()
    ctx.stage_reset()
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    n_kmers = ctx.n_kmers
    stages = ctx.stage_ms()
    lk_ms, lk_n, lk_q = ctx.lookup_kernel_ms()
    ca, cu, _ = ctx.profile()
    identified = float(ca[-1].sum()) / max(1, args.steps) / max(1, n_kmers)

    if rank == 0:
        total_reads = args.reads * world * args.steps
        value = total_reads / dt
        # roofline kernel: lookup_tile_kernel (the sorted-index lookup BASELINE.json's 40 % target names).  Algorithmic
        # bytes per launch: every sorted query record once (8 B key + 4 B read id) + every
        # index record once (12 B).  The other stages are listed with their own times in stage_ms_per_step.
        algo_bytes = n_kmers * rec_bytes + ix.n * rec_bytes
        lk_avg_s = (lk_ms / max(1, lk_n)) / 1e3
        achieved = algo_bytes / lk_avg_s / 1e9 if lk_avg_s > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_lookup_pmc.json")
        if os.path.exists(pmc) and not args.wide and args.reads == 10_000_000:   # measured for that launch only
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "reads/s in identify (10M x 150bp vs k=12 index)" if not args.wide else
                      "reads/s in identify (150bp reads vs k<=25 128-bit index)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u128" if args.wide else "u64",
            "data": "synthetic",
            "config": {"workload": f"{args.reads} synthetic {args.read_len} bp reads per GPU vs {ix.n}-record "
                                   f"({ix.n * rec_bytes / 1e9:.1f} GB) "
                                   + ("k<=25 128-bit index, -k 25 7, 3 frames, " if args.wide else "k<=12 64-bit index, -k 12 7, 3 frames, ") +
                                   f"{'profile only' if args.profile_only else 'profile + per-read scores'}",
                       "reads_per_gpu": args.reads, "kmers_per_gpu": n_kmers, "index_records": int(ix.n),
                       "taxa": args.taxa, "parallelism": f"read-sharded x{world}, index replicated"},
            "kmers_per_s": n_kmers * world * args.steps / dt,
            "identified_fraction": identified,
            "reads_on_general_score_kernel": ctx.last_slow_reads(),
            "stage_ms_per_step": {k: v[0] / max(1, args.steps) for k, v in stages.items()},
            # minimum HBM traffic of a stage / its measured time (GB/s).  encode: bases in +
            # (key, read id) out; sort: one read + one write of the records (the 8 radix passes it really takes are the
            # implementation's); lookup: as `roofline` but over the whole stage; group: key + depth + rep in, nK x 8 B out
            "stage_algorithmic_gbps": stage_gbps(stages, args.steps, n_kmers, args.reads * args.read_len, ix.n,
                                                rec_bytes, k_high - k_low + 1),
            "roofline": {"bound": "hbm", "kernel": "lookup_tile_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": lk_avg_s * 1e3},
        }
        if not args.no_cpu and world == 1:                # the CPU baseline is reported at N = 1 only
            sample = reads.slice(0, min(args.cpu_sample, reads.n))
            v, secs = cpu_baseline(ix, sample, k_high, k_low)
            out["cpu_baseline"] = {"value": v, "unit": "reads/s", "cores": 1, "kind": "port",
                                   "sample": f"first {sample.n} reads of the same workload, same index, "
                                             f"oracle/ (C restatement of the reference, 1 thread), {secs:.1f} s"}
        print(json.dumps(out), flush=True)
    ctx.close()
    dix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
