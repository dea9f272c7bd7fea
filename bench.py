#!/usr/bin/env python3
"""bench.py -- reads/s of the `identify` hot path on MI355X (BASELINE.json metric).

One step = one pass of the whole hot path (encode -> sort -> lookup -> group -> score, per-read CSR included) over
one batch of synthetic reads that is already resident in HBM.  Workload at N=1: BASELINE.json configs[1]: 10 M
synthetic 150 bp reads against a ~5 GB k<=12 64-bit index (1400 taxa x 300 kb, sibling genomes 3 % apart; 1 % read
errors; -k 12 7, three frames).  With --gpus N every rank holds the whole index and its own 10 M reads (weak scaling,
BASELINE.json configs[3]); the per-rank profile tables are summed with one RCCL all-reduce per step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--taxa G] [--genome-len L]

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, before anything touches a
GPU; under `python -m torch.distributed.run` it takes RANK / LOCAL_RANK / WORLD_SIZE from the environment.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(ix, sample, k_high, k_low, threads):
    """The CPU oracle (a port of the reference algorithm) on a bounded sample, with the reference's threading model:
    -n worker threads over read chunks (encode) and range-aligned slices (lookup + score), oracle/kasa_oracle.c."""
    from oracle import oracle
    p = oracle.params(k_high, k_low, 3, K=ix.K)
    iv = oracle.IndexView(ix)
    t0 = time.perf_counter()
    oracle.identify_threaded(iv, sample.bases, sample.offsets, p, threads)
    dt = time.perf_counter() - t0
    return sample.n / dt, dt


def kernel_bytes(n_q, n_idx, rec_bytes, rec_words, stats):
    """Algorithmic HBM bytes per launch of the individually timed kernels."""
    key = rec_bytes - 4
    rec = 4 * rec_words
    return {
        # every sorted query record once + every index record once (the merge-join lower bound)
        "lookup_tile_kernel": n_q * rec_bytes + n_idx * rec_bytes,
        # key + depth + index position + slot in, one event record out per query; meta + taxon of every index record
        "group_kernel": n_q * (key + 1 + 4 + 4 + rec) + n_idx * 5,
        # every event record once; per read two offsets in, 16 bytes out
        "score_main_kernel": n_q * rec,
        # every event record once, 8 bytes per staging record out
        "score_other_kernel": n_q * rec + 8 * stats["staging_records"],
        # staging records twice (bitmap pass, replay pass), final rows and profile keys out
        "row_merge_kernel": 16 * stats["staging_records"] + 8 * stats["nnz"] + 8 * stats["profile_keys"],
    }


def stage_bytes(n_q, n_bases, n_idx, rec_bytes, rec_words, stats):
    """minimum HBM traffic of a stage.  encode: bases in + (key, payload) out; sort: one read + one
    write of the records (the radix passes it really takes are the implementation's); lookup: every query record and
    every index record once; group: as the kernel; score: every event record twice (main chains, other taxa) + the
    staging records written and read twice + profile keys written, sorted once (read + write) and reduced + the CSR out --
    the per-(event, taxon) contributions section 8(d) counts travel inside those records."""
    kb = kernel_bytes(n_q, n_idx, rec_bytes, rec_words, stats)
    return {"encode": n_bases + n_q * rec_bytes, "sort": 2 * n_q * rec_bytes, "lookup": kb["lookup_tile_kernel"],
            "group": kb["group_kernel"],
            "score": kb["score_main_kernel"] + kb["score_other_kernel"] + kb["row_merge_kernel"] + 32 * stats["profile_keys"] + 16 * stats["nnz"]}


def launch_ranks(n):
    """Start the n ranks of a multi-GPU run ourselves: fresh processes, env set before anything touches a GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is collected by a reader thread; the launcher polls: a rank that dies (OOM, RCCL init) would leave the
    # others blocked in a collective for ever -- then the rest is ended and its exit code returned
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("KASA_BENCH_TIMEOUT_S", "3000"))
    failed = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            failed = abs(bad[0]) if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    return failed


def measure(args, ctx, reads, ix, world, dist, share, wide, torch, kdist):
    """Warm up, time exactly --steps steps between barriers; -> dict of raw measurements."""
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
    return dt


def report(args, ctx, reads, ix, world, dt, wide, pcie):
    k_high, k_low = (25, 7) if wide else (12, 7)
    rec_bytes = 20 if wide else 12
    n_kmers = ctx.n_kmers
    stages = ctx.stage_ms()
    kern = ctx.kernel_ms()
    stats = ctx.batch_stats()
    ca, cu, _ = ctx.profile()
    identified = float(ca[-1].sum()) / max(1, args.steps) / max(1, n_kmers)
    n_reads = reads.n
    value = n_reads * world * args.steps / dt
    kb = kernel_bytes(n_kmers, ix.n, rec_bytes, ctx.rec_words, stats)
    kernels = {}
    for name, (ms, n) in kern.items():
        if n:
            avg = ms / n
            kernels[name] = {"avg_launch_ms": avg, "algorithmic_bytes_per_launch": kb[name],
                             "achieved": kb[name] / (avg * 1e-3) / 1e9, "frac": kb[name] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # the roofline line is the kernel with the largest share of the step
    dom = max(kernels, key=lambda k: kernels[k]["avg_launch_ms"]) if kernels else None
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r02_kernel_pmc.json")
    if dom and os.path.exists(pmc) and not wide and n_reads == 10_000_000:   # measured for that workload only
        try:
            traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    sb = stage_bytes(n_kmers, n_reads * args.read_len, ix.n, rec_bytes, ctx.rec_words, stats)
    out = {
        "metric": "reads/s in identify (10M x 150bp vs k=12 index)" if not wide else
                  "reads/s in identify (150bp reads vs k<=25 128-bit index)", "value": value, "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u128" if wide else "u64",
        "data": "synthetic",
        "config": {"workload": f"{n_reads} synthetic {args.read_len} bp reads per GPU vs {ix.n}-record "
                               f"({ix.n * rec_bytes / 1e9:.1f} GB) "
                               + ("k<=25 128-bit index, -k 25 7, 3 frames, " if wide else "k<=12 64-bit index, -k 12 7, 3 frames, ") +
                               f"{'profile only' if args.profile_only else 'profile + per-read scores'}",
                   "reads_per_gpu": n_reads, "kmers_per_gpu": n_kmers, "index_records": int(ix.n),
                   "taxa": args.taxa, "parallelism": f"read-sharded x{world}, index replicated"},
        "kmers_per_s": n_kmers * world * args.steps / dt,
        "identified_fraction": identified,
        "batch": stats,
        "stage_ms_per_step": {k: v[0] / max(1, args.steps) for k, v in stages.items()},
        "stage_algorithmic_gbps": {k: (b / (stages[k][0] / max(1, args.steps) * 1e-3) / 1e9 if stages.get(k, (0,))[0] > 0 else None)
                                   for k, b in sb.items()},
        "roofline": dict({"bound": "hbm", "kernel": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic},
                         **(kernels.get(dom, {}))),
        "kernels": kernels,
    }
    if pcie is not None:
        out["e2e"] = pcie
    return out


def pcie_inclusive(ctx, reads, want, ix, k_high):
    """Two more passes with the PCIe legs inside the clock (never `value`): host reads in, and out either what the per-read
    file can print (ranked on the device, kasa_batch_rank; page-locked buffers) or the whole CSR (pageable memory, the
    round-1 path).  The file-to-file rate of the C++ driver is measured by tools/e2e_host.py."""
    from kasa_amd import capi, report
    import numpy as np
    out = {}
    if want:
        bases = capi.pinned_empty(reads.bases.shape[0], np.uint8)
        offsets = capi.pinned_empty(reads.offsets.shape[0], np.int64)
        bases[:] = reads.bases
        offsets[:] = reads.offsets
        den, rclass = report.rank_denominators(ix.freq_at(k_high), reads.lengths, ix.K, False)
        nnz = int(ctx.batch_stats()["nnz"])                                # a long-lived host keeps its page-locked buffers
        csr_buf = (capi.pinned_empty(reads.n + 1, np.uint64), capi.pinned_empty(nnz, np.uint32), capi.pinned_empty(nnz, np.float32))
        rank_buf = (capi.pinned_empty(reads.n * 4, np.uint32), capi.pinned_empty(reads.n * 8, capi.RANK_ENTRY))
        ctx.rank(den, rclass, 0.0, 3, out=rank_buf)                          # ... and the context its device buffers (the batch is scored)
        t0 = time.perf_counter()
        ctx.upload(bases, offsets)
        ctx.encode()
        ctx.sort_and_range()
        ctx.lookup_score(True, False)
        ctx.synchronize()
        t1 = time.perf_counter()
        meta, ent, flagged = ctx.rank(den, rclass, 0.0, 3, out=rank_buf)
        t2 = time.perf_counter()
        nbytes = int(meta.nbytes + ent.nbytes)
        if flagged:          # reads with tied hits (all synthetic taxa have the same frequency): the host ranks them from the full rows
            csr = ctx.scores(out=csr_buf)
            nbytes += int(sum(a.nbytes for a in csr))
        ctx.profile()
        dt = time.perf_counter() - t0
        out.update({"pcie_inclusive_reads_per_s": reads.n / dt, "pcie_inclusive_s_per_batch": dt,
                    "ranked_entries": int(ent.shape[0]), "reads_ranked_by_host": int(flagged), "downloaded_bytes": nbytes,
                    "upload_and_device_s": t1 - t0, "rank_and_fetch_s": t2 - t1, "csr_and_profile_s": t0 + dt - t2})
    t0 = time.perf_counter()
    ctx.upload(reads.bases, reads.offsets)
    ctx.encode()
    ctx.sort_and_range()
    ctx.lookup_score(want, False)
    if want:
        ctx.scores()
    ctx.profile()
    dt = time.perf_counter() - t0
    out.update({"csr_download_reads_per_s": reads.n / dt, "csr_download_s_per_batch": dt,
                "note": "pcie_inclusive: page-locked reads up, device, ranking on the device (-b 3), ranked hits + profile down "
                        "(+ the CSR into page-locked memory when the device hands reads back); "
                        "csr_download: the same with the whole CSR down into pageable memory; file to file: tools/e2e_host.py"})
    if not want:
        out["pcie_inclusive_reads_per_s"] = out["csr_download_reads_per_s"]
        out["pcie_inclusive_s_per_batch"] = out["csr_download_s_per_batch"]
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
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive extra pass")
    ap.add_argument("--profile-only", action="store_true", help="no per-read scores (kASA without -q)")
    ap.add_argument("--wide", action="store_true",
                    help="BASELINE.json configs[2] as the only measurement: 128-bit index, -k 25 7")
    ap.add_argument("--secondary", action="store_true",
                    help="after the headline measurement also run configs[2] (128-bit index, -k 25 7, same reads) and report it "
                         "as `secondary` inside the one JSON line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))      # nothing has touched a GPU in this process

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (one-GPU boxes): KASA_BENCH_SHARE_GPU=1 runs every rank on device 0 and carries the reduce over gloo,
    # because RCCL refuses two ranks on one device; it exercises the multi-rank code path, it is not a measurement
    share = os.environ.get("KASA_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

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

    def one(wide):
        k_high, k_low = (25, 7) if wide else (12, 7)
        rec_bytes = 20 if wide else 12
        t0 = time.perf_counter()
        g = synth.genomes(args.taxa, args.genome_len, seed=11)
        ix = synth.index_from_genomes(g, device=local_rank, K=25 if wide else 12)
        log(f"[rank {rank}] index: {ix.n} records ({ix.n * rec_bytes / 1e9:.2f} GB on disk layout), "
            f"{ix.trie_prefix.shape[0]} prefixes, {time.perf_counter() - t0:.1f} s")
        t0 = time.perf_counter()
        reads = synth.reads_from_genomes(g, args.reads, args.read_len, seed=1000 + rank)
        log(f"[rank {rank}] reads: {reads.n} x {args.read_len} bp, {time.perf_counter() - t0:.1f} s")
        dix = capi.DeviceIndex(ix, local_rank, check_trie=True)
        ctx = capi.Context(dix, k_high, k_low, 3)
        ctx.upload(reads.bases, reads.offsets)         # inputs resident in HBM before the timed region
        dt = measure(args, ctx, reads, ix, world, dist, share, wide, torch, kdist)
        pcie = None
        out = None
        if rank == 0:
            if world == 1 and not args.no_e2e:
                out0 = report(args, ctx, reads, ix, world, dt, wide, None)     # (stats of the timed run, before the extra pass)
                try:                                               # an extra pass: it must never cost the headline line
                    pcie = pcie_inclusive(ctx, reads, not args.profile_only, ix, k_high)
                except Exception as ex:                            # e.g. no memory left for the page-locked buffers
                    pcie = {"error": str(ex)[:300]}
                out0["e2e"] = pcie
                out = out0
            else:
                out = report(args, ctx, reads, ix, world, dt, wide, None)
            if not args.no_cpu and world == 1 and not wide:   # the CPU baseline is reported at N = 1 only
                threads = os.cpu_count() or 1
                sample = reads.slice(0, min(args.cpu_sample, reads.n))
                v1, s1 = cpu_baseline(ix, sample, k_high, k_low, 1)
                vn, sn = cpu_baseline(ix, sample, k_high, k_low, threads)
                cal = {}
                try:
                    cal = json.load(open(os.path.join(ROOT, "profiles", "cpu_calibration.json")))
                except Exception:
                    pass
                out["cpu_baseline"] = {"value": vn, "unit": "reads/s", "cores": threads, "kind": "port",
                                       "cpu": cpu_model(), "single_thread_value": v1,
                                       "sample": f"first {sample.n} reads of the same workload, same index, oracle/ (C restatement of "
                                                 f"the reference with its threading model), {sn:.1f} s with {threads} threads, {s1:.1f} s with 1",
                                       "calibration": cal}
        ctx.close()
        dix.close()
        return out

    out = one(args.wide)
    if args.secondary and not args.wide:
        sec = one(True)
        if rank == 0 and out is not None and sec is not None:
            out["secondary"] = {k: sec[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "kmers_per_s",
                                                    "identified_fraction", "batch", "stage_ms_per_step", "roofline", "kernels")}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
