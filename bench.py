#!/usr/bin/env python3
"""bench.py -- reads/s of the `identify` hot path on MI355X (BASELINE.json metric).

One step, AT EVERY N, is the same loop: for each of the rank's batches (bases and offsets already resident in HBM)
kasa_batch_upload_device (the read geometry, no PCIe) -> encode -> sort -> lookup -> group -> score (per-read CSR
included); with N > 1 the step ends with ONE RCCL all-reduce of the profile tables through the C ABI
(kasa_profile_allreduce on the context's stream; its own communicator, created from an ncclUniqueId that travels over
torch.distributed).  `value` = all reads of all ranks / max-over-ranks time.

  N = 1   BASELINE.json configs[1] (C2): ONE batch of 10 M synthetic 150 bp reads against a ~5 GB k<=12 64-bit index
          (1400 taxa x 300 kb, sibling genomes 3 % apart; 1 % read errors; -k 12 7, three frames).  The same line also
          carries `secondary` = configs[2] (C3: the same reads against the 128-bit index, -k 25 7), `tertiary` (a crowded
          index: clades of taxa sharing conserved genes), `quaternary` (long reads: 100 000 x 10 kb reads and two 9.6 Mbp
          contigs against the same index), `e2e` (PCIe-inclusive rates and the file-to-file rate of the
          C++ driver: FASTQ in, JSONL + profile out, index load excluded), `cpu_baseline` (the oracle with the
          reference's threading model on the host cores) and, as its LAST key, `summary` (every leg in under 1900
          characters: a reader that keeps the line's tail sees them all).  At N = 1 the process imports no torch: the reads
          lie in plain device buffers of the C ABI and the library runs on the HIP runtime it was built for (`runtime`).
  N > 1   the same 10 M-read batch on EVERY rank, index replicated ("weak": N x 10 M reads per step; N = 1 and N = 8 differ
          by the reduce only).  The same line carries `c2_strong` = BASELINE.json's metric read literally (10 M reads in all,
          10 M / N per rank) and `c4` = BASELINE.json configs[3]: 100 M reads in all, 100 M / N per rank in batches of at
          most 10 M (both "strong").  `--total-reads T` makes that form the headline instead.
  --partitioned   BASELINE.json configs[4] (C5): the index range-partitioned over the ranks (kasa_amd/dist.py), one
          slice per rank made of the genomes' k-mers of its prefix range plus random filler records.

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


DEVICE = [0]                                           # the rank's device (main sets it)


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


def cpu_baseline_report(ix, reads, k_high, k_low, args):
    """One thread on a small sample, all cores on a large one (bounded by the host's free memory: the oracle keeps the
    reference's dense reads x taxa score matrix), and the speed-up between the two."""
    host = host_cpus()
    threads = host["usable"]
    per_read = ix.content.n_taxa * 4 + 130 * 40 + 400                    # score row + query records (two copies, ranges) + text
    avail = None
    try:
        import psutil
        avail = int(psutil.virtual_memory().available)
    except Exception:
        pass
    n_par = min(reads.n, args.cpu_sample_parallel)
    if avail is not None:
        n_par = max(min(n_par, int(0.25 * avail / per_read)), min(reads.n, args.cpu_sample))
    if threads < 16:
        n_par = min(n_par, args.cpu_sample)
    one = reads.slice(0, min(args.cpu_sample, reads.n))
    v1, s1 = cpu_baseline(ix, one, k_high, k_low, 1)
    par = reads.slice(0, n_par)
    vn, sn = cpu_baseline(ix, par, k_high, k_low, threads)
    cal = {}
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "cpu_calibration.json")))
    except Exception:
        pass
    return {"value": vn, "unit": "reads/s", "cores": threads, "threads": threads, "kind": "port", "cpu": cpu_model(), "host_cpus": host,
            "single_thread_value": v1, "speedup_over_1": vn / v1 if v1 > 0 else None,
            "parallel_efficiency": (vn / v1 / threads) if v1 > 0 else None,
            "sample": f"first {par.n} reads of the same workload with {threads} threads ({sn:.1f} s), first {one.n} reads with one "
                      f"thread ({s1:.1f} s); same index; oracle/ = C restatement of the reference with its threading model "
                      "(reads split for the translation, parallel sort, range-aligned slices merged into the shared dense "
                      "score matrix, private count tables)",
            "host_memory_available_gb": None if avail is None else avail / 1e9,
            "calibration": cal}


def kernel_bytes(n_q, n_idx, rec_bytes, rec_words, stats):
    """Algorithmic HBM bytes per launch of the individually timed kernels."""
    key = rec_bytes - 4
    rec = 4 * rec_words
    return {
        # every sorted query record once + every index record once (the merge-join lower bound)
        "lookup_tile_kernel": n_q * rec_bytes + n_idx * rec_bytes,
        # key + depth + index position + slot in, one event record out per query; meta + taxon of every index record; the profile keys out
        "group_kernel": n_q * (key + 1 + 4 + 4 + rec) + n_idx * 5 + 8 * stats["profile_keys"],
        # every event record once; per read two offsets in, 16 bytes out
        "score_main_kernel": n_q * rec,
        # every event record once, 8 bytes per staging record out
        "score_other_kernel": n_q * rec + 8 * stats["staging_records"],
        # staging records twice (bitmap pass, replay pass), final rows out
        "row_merge_kernel": 16 * stats["staging_records"] + 8 * stats["nnz"],
    }


def stage_bytes(n_q, n_bases, n_idx, rec_bytes, rec_words, stats):
    """minimum HBM traffic of a stage.  encode: bases in + (key, payload) out; sort: one read + one
    write of the records (the radix passes it really takes are the implementation's); lookup: every query record and
    every index record once; group: as the kernel + the profile keys (one per leader, segment and level) read once by the
    table kernel; score: every event record twice (main chains, other taxa) + the staging records written and read twice +
    the CSR out -- the per-(event, taxon) contributions section 8(d) counts travel inside those records."""
    kb = kernel_bytes(n_q, n_idx, rec_bytes, rec_words, stats)
    return {"encode": n_bases + n_q * rec_bytes, "sort": 2 * n_q * rec_bytes, "lookup": kb["lookup_tile_kernel"],
            "group": kb["group_kernel"] + 8 * stats["profile_keys"],
            "score": kb["score_main_kernel"] + kb["score_other_kernel"] + kb["row_merge_kernel"] + 16 * stats["nnz"]}


def run_with_retry():
    """The default line is measured by a child process (this one has not touched a GPU).  If the child ends without its line
    it is measured ONCE more -- and the line says so: `attempts`, `retried`, `first_rc` (negative: the signal) and the tail of
    the failed child's stderr travel in the JSON (round 4 retried silently, which hid an unexplained crash from the record)."""
    import tempfile
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--child"]
    rc, first = 1, None
    for attempt in (1, 2):
        with tempfile.TemporaryFile(mode="w+") as err:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=err, text=True)
            err.seek(0)
            tail = err.read()
        sys.stderr.write(tail)
        sys.stderr.flush()
        lines = [ln for ln in (r.stdout or "").splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            try:
                out = json.loads(lines[-1])
                out["attempts"] = attempt
                out["retried"] = attempt > 1
                if first is not None:
                    out["first_rc"], out["first_stderr_tail"] = first
                print(json.dumps(out), flush=True)
            except ValueError:
                sys.stdout.write(r.stdout)
                sys.stdout.flush()
            return 0
        log(f"bench.py: the measuring process ended with code {r.returncode} and {'a' if lines else 'no'} line (attempt {attempt} of 2)")
        if first is None:
            first = (r.returncode, tail[-1500:])
        rc = r.returncode or 1
    return rc


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


def resident_batches(torch, synth, g, per_rank, max_batch, L, seed0):
    """The rank's reads in HBM: one uint8 tensor + per batch a device tensor of offsets.  -> (tensors to keep alive,
    [(bases pointer, offsets pointer, reads)], the first part as a host ReadBatch for the report and the e2e legs)."""
    n_batches = max(1, -(-per_rank // max_batch))
    per_batch = -(-per_rank // n_batches)
    if torch is None:                                                      # one GPU: plain device buffers of the C ABI, no other GPU library in the process
        import numpy as np
        from kasa_amd import capi
        dev = capi.DeviceBuffer(per_rank * L, DEVICE[0])
        first, done = None, 0
        while done < per_rank:
            m = min(max_batch, per_rank - done)
            part = synth.reads_from_genomes(g, m, L, seed=seed0 + done // max(1, max_batch))
            dev.write(part.bases, done * L)
            first = first or part
            done += m
        keep, batches = [dev], []
        for b in range(n_batches):
            a0, a1 = b * per_batch, min(per_rank, (b + 1) * per_batch)
            off = capi.DeviceBuffer((a1 - a0 + 1) * 8, DEVICE[0])
            off.write(np.arange(a1 - a0 + 1, dtype=np.int64) * L)
            keep.append(off)
            batches.append((dev.ptr + a0 * L, off.ptr, a1 - a0))
        return keep, batches, first
    dev_reads = torch.empty(per_rank * L, dtype=torch.uint8, device="cuda")
    first = None
    done = 0
    while done < per_rank:
        m = min(max_batch, per_rank - done)
        part = synth.reads_from_genomes(g, m, L, seed=seed0 + done // max(1, max_batch))
        dev_reads[done * L:(done + m) * L].copy_(torch.from_numpy(part.bases))
        if first is None:
            first = part
        done += m
    keep, batches = [dev_reads], []
    for b in range(n_batches):
        a0, a1 = b * per_batch, min(per_rank, (b + 1) * per_batch)
        off = torch.arange(a1 - a0 + 1, dtype=torch.int64, device="cuda") * L
        keep.append(off)
        batches.append((dev_reads.data_ptr() + a0 * L, off.data_ptr(), a1 - a0))
    torch.cuda.synchronize()
    return keep, batches, first


def measure(args, ctx, world, dist, share, torch, kdist, batches, comm=0):
    """Warm up, time exactly --steps steps between barriers.  batches = [(device pointer of the bases, device pointer of the
    offsets, reads)]: the rank's reads, resident in HBM, taken batch by batch -- the same loop at every N; with N > 1 the step
    ends with the profile reduce over the ranks.  -> {"dt": seconds (max over ranks), per-rank times, reduce / upload time}."""
    want = not args.profile_only
    acc = {"reduce_s": 0.0}

    def reduce_profile():
        if dist is None or world == 1:
            return
        ctx.synchronize()                                                  # (the step's device work is done: what follows is the reduce alone,
        t0 = time.perf_counter()                                           #  waiting for the slowest rank included)
        if comm:
            ctx.profile_allreduce(comm)                                    # C ABI: pack on the device, ncclAllReduce on the context's stream, unpack
        else:
            ctx.profile_set_limbs(kdist.allreduce_limbs(ctx.profile_limbs(), device=None if share else "cuda"))   # (shared-GPU test hook / fallback: through torch.distributed)
        ctx.synchronize()
        acc["reduce_s"] += time.perf_counter() - t0

    def step():
        ctx.profile_reset()                                                # a step is a whole "file": its own profile, summed over the ranks at its end
        for bases_ptr, off_ptr, n in batches:
            ctx.upload_resident(bases_ptr, off_ptr, n)
            ctx.encode()
            ctx.sort_and_range()
            ctx.lookup_score(want, False)
        reduce_profile()

    def fence():
        ctx.synchronize()
        if torch is not None:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if torch is not None:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # what the per-batch upload costs (geometry kernels, two scans, one 8-byte read-back), timed alone before the run
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.upload_resident(*batches[0])
    ctx.synchronize()
    upload_ms = (time.perf_counter() - t0) / 3 * 1e3
    fence()
    ctx.stage_reset()
    acc["reduce_s"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    mine = time.perf_counter() - t0                                        # this rank's own time, before it waits for the others
    fence()
    dt = time.perf_counter() - t0
    res = {"dt": dt, "rank_s": [mine], "reduce_ms_per_step": acc["reduce_s"] / max(1, args.steps) * 1e3}
    if dist is not None and world > 1:
        dev = "cpu" if share else "cuda"
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        res["dt"] = float(tmax.item())
        every = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(every, torch.tensor([mine, acc["reduce_s"]], dtype=torch.float64, device=dev))
        res["rank_s"] = [float(t[0].item()) for t in every]
        res["reduce_ms_per_step"] = max(float(t[1].item()) for t in every) / max(1, args.steps) * 1e3
    res["upload_ms_per_batch"] = upload_ms
    return res


def source_sha16():
    """What the committed PMC / profile files are stamped with: a hash of the kernel sources (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("kasa_hip.hip", "kasa_radix.h", "kasa_text.h", "stdsort_order.h"):
        with open(os.path.join(ROOT, "kasa_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def scatter_bound(n_q, avg_ms, rec_words):
    """group_kernel ends in one random record store per query; tools/scatter_probe.hip measured what this chip takes
    (profiles/r04_scatter_probe.json): the second bound of the roofline object."""
    try:
        rows = json.load(open(os.path.join(ROOT, "profiles", "r04_scatter_probe.json")))["rows"]
    except Exception:
        return None
    want = ("scatter, whole buffer, 256-thread workgroups", 32) if rec_words == 8 else \
           ("scatter of 64-byte records, lane per record, stored by quads through LDS (4 instructions of 16 records)", 64)
    peak = [r["g_records_per_s"] for r in rows if r["what"] == want[0] and r["record_bytes"] == want[1]]
    if not peak or not avg_ms:
        return None
    achieved = n_q / (avg_ms * 1e-3) / 1e9
    return {"bound": "scatter", "kernel": "group_kernel", "peak": peak[0], "unit": "G records/s", "achieved": achieved, "frac": achieved / peak[0],
            "source": "profiles/r04_scatter_probe.json: \"%s\", %d-byte records into a 34 GB buffer" % want}


def report(args, ctx, reads, ix, world, res, wide, n_reads, n_batches, scaling, extra_cfg=None, traffic=None):
    """n_reads: reads of one rank per step (in n_batches batches); res: measure()'s times."""
    k_high, k_low = (25, 7) if wide else (12, 7)
    rec_bytes = 20 if wide else 12
    dt = res["dt"]
    n_kmers = ctx.n_kmers                                                 # of the last batch: what the per-kernel byte counts refer to
    stages = ctx.stage_ms()
    kern = ctx.kernel_ms()
    stats = ctx.batch_stats()
    ca, cu, _ = ctx.profile()
    kmers_step = n_reads * max(0, args.read_len - 3 * k_low + 1)          # windows of a read: len - 3 kLow + 1 (Read.hpp:36-57,1074-1077)
    identified = float(ca[-1].sum()) / max(1, world) / max(1, kmers_step)
    value = n_reads * world * args.steps / dt
    kb = kernel_bytes(n_kmers, ix.n, rec_bytes, ctx.rec_words, stats)
    kernels = {}
    for name, (ms, n) in kern.items():
        if n and name in kb:
            avg = ms / n
            kernels[name] = {"avg_launch_ms": avg, "algorithmic_bytes_per_launch": kb[name],
                             "achieved": kb[name] / (avg * 1e-3) / 1e9, "frac": kb[name] / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS}
        elif n:                                                            # groups of kernels timed together (no single byte count): ms per step
            kernels[name] = {"ms_per_step": ms / max(1, args.steps), "timed_sections_per_step": n / max(1, args.steps)}
    # the roofline line is the single kernel with the largest share of the step
    single = [k for k in kernels if "avg_launch_ms" in kernels[k]]
    dom = max(single, key=lambda k: kernels[k]["avg_launch_ms"]) if single else None
    per_batch_reads = n_reads // max(1, n_batches)
    sb = stage_bytes(n_kmers, per_batch_reads * args.read_len, ix.n, rec_bytes, ctx.rec_words, stats)
    sb = {k: v * n_batches for k, v in sb.items()}
    roof = dict({"bound": "hbm", "kernel": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}, **(kernels.get(dom, {})))
    if "group_kernel" in kernels:
        sc = scatter_bound(n_kmers, kernels["group_kernel"]["avg_launch_ms"], ctx.rec_words)
        if sc:
            roof["second_bound"] = sc
    rank_ms = [t / args.steps * 1e3 for t in res["rank_s"]]
    out = {
        "metric": "reads/s in identify (10M x 150bp vs k=12 index)" if not wide else
                  "reads/s in identify (150bp reads vs k<=25 128-bit index)", "value": value, "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u128" if wide else "u64",
        "data": "synthetic",
        "config": dict({"workload": (f"{n_reads} synthetic {args.read_len} bp reads per GPU" if world == 1 else
                                     f"{n_reads * world} synthetic {args.read_len} bp reads in all, {n_reads} per GPU") +
                                    f" in {n_batches} batch{'es' if n_batches > 1 else ''} resident in HBM" +
                                    f" vs {ix.n}-record ({ix.n * rec_bytes / 1e9:.1f} GB) "
                                    + ("k<=25 128-bit index, -k 25 7, 3 frames, " if wide else "k<=12 64-bit index, -k 12 7, 3 frames, ") +
                                    f"{'profile only' if args.profile_only else 'profile + per-read scores'}",
                        "step": "per batch: kasa_batch_upload_device (read geometry on the device, no PCIe) + encode + sort + lookup + group + score"
                                + ("; then one all-reduce of the profile tables" if world > 1 else ""),
                        "reads_per_gpu": n_reads, "batches_per_step": n_batches, "kmers_per_gpu": kmers_step, "index_records": int(ix.n),
                        "taxa": args.taxa, "parallelism": f"read-sharded x{world}, index replicated"}, **(extra_cfg or {})),
        "kmers_per_s": kmers_step * world * args.steps / dt,
        "identified_fraction": identified,
        "rank_step_ms": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
        "reduce_ms_per_step": res["reduce_ms_per_step"], "upload_ms_per_batch": res["upload_ms_per_batch"],
        "batch": stats,
        "record_placement": ctx.record_placement(),                       # how the record buffer was chosen: the rate of random 32-byte stores is a property of the memory behind it (tools/place_probe.hip)
        "stage_ms_per_step": {k: v[0] / max(1, args.steps) for k, v in stages.items()},
        "stage_algorithmic_gbps": {k: (b / (stages[k][0] / max(1, args.steps) * 1e-3) / 1e9 if stages.get(k, (0,))[0] > 0 else None)
                                   for k, b in sb.items()},
        "roofline": roof,
        "kernels": kernels,
    }
    return out


def pmc_traffic(args, wide, kernel, live=True, crowded=False):
    """HBM bytes per launch of `kernel`, measured NOW: two child runs of this file under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE: separate passes, counters only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 on gfx950), one step of the
    same workload each.  The caller has released its device memory.  Falls back to the committed passes of the same kernel
    sources (profiles/r04_kernel_pmc.json, stamped with source_sha16) and says which."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    want = {"score_other_kernel": "score_other_flat", "row_merge_kernel": "row_merge_bitmap_kernel", "group_kernel": "group2?_kernel"}.get(kernel, kernel)   # (regular expressions)
    got = {}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if live and os.path.exists(exe):
        d = tempfile.mkdtemp(prefix="kasa_pmc_")
        try:
            for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_SALU")):
                counter = counters[0]
                dd = os.path.join(d, counter)
                cmd = [exe, "--pmc", *counters, "--kernel-include-regex", want, "--output-format", "csv", "-d", dd, "--", sys.executable,
                       os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e", "--no-secondary", "--no-tertiary", "--no-quaternary", "--no-pmc",
                       "--reads", str(args.reads), "--taxa", str(args.taxa), "--genome-len", str(args.genome_len), "--read-len", str(args.read_len)]
                if wide:
                    cmd.append("--wide")
                if crowded:
                    cmd += ["--crowded", "--crowded-reads", str(args.crowded_reads), "--warmup", "1"]   # (the first step sizes its buffers; the largest dispatch is taken)
                env = dict(os.environ, TMPDIR="/tmp")
                for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                    env.pop(k, None)
                subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, cwd="/tmp")
                vals = {c: [] for c in counters}
                for f in glob.glob(dd + "/**/*counter_collection.csv", recursive=True):
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] in vals and re.search(want, row["Kernel_Name"]):
                            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                for c, v in vals.items():
                    if v:
                        got[c] = max(v)                                    # the largest dispatch: the batch (the index build launches some kernels too)
        except Exception as ex:
            got["error"] = str(ex)[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
        out = {"traffic": got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024,
               "traffic_source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two child passes of one step each; FETCH_SIZE x 2, gfx950)"}
        if "SQ_INSTS_VALU" in got and "SQ_INSTS_SALU" in got:
            out["insts"] = {"valu": got["SQ_INSTS_VALU"], "salu": got["SQ_INSTS_SALU"],
                            "source": "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU (a third child pass of one step)"}
        return out
    if crowded:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r05_kernel_pmc_crowded.json")))
            e = pmc.get(kernel, {})
            if pmc.get("source_sha16") == source_sha16() and e.get("hbm_bytes_per_launch"):
                return {"traffic": e["hbm_bytes_per_launch"], "traffic_source": "profiles/r05_kernel_pmc_crowded.json (same kernel sources, source_sha16 %s)" % pmc["source_sha16"]}
        except Exception:
            pass
        return {"traffic": None, "traffic_source": "none: rocprofv3 pass failed here (%s) and no committed pass of these kernel sources" % got.get("error", "no counters")}
    if (args.reads, args.taxa, args.genome_len, args.read_len) != (10_000_000, 1400, 300_000, 150):
        return {"traffic": None, "traffic_source": "none: the committed counter passes are of the default workload"}
    try:
        name = "r05_kernel_pmc%s.json" % ("_wide" if wide else "")
        pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        if pmc.get("source_sha16") != source_sha16():
            return {"traffic": None, "traffic_source": "none: rocprofv3 pass failed here (%s) and profiles/%s was taken from other kernel sources" % (got.get("error", "no counters"), name)}
        e = pmc.get("group2_kernel" if kernel == "group_kernel" and "group2_kernel" in pmc else kernel, {})
        return {"traffic": e.get("hbm_bytes_per_launch"),
                "traffic_source": "profiles/%s (same kernel sources, source_sha16 %s)" % (name, pmc["source_sha16"])}
    except Exception:
        return {"traffic": None, "traffic_source": "none"}


def pcie_pipelined(ctx, dix, reads, bases, offsets, den, rclass, csr_buf, rank_buf, k_high, k_low, per_ctx=3):
    """The long-lived host of INTEGRATION.md section 2b: TWO contexts on one device index, one host thread each, every
    context with its own stream and page-locked buffers -- the upload of one batch and the ranking + download of another
    run beside the kernels of a third (the reference has its own output thread for the same reason, Compare.hpp:3390-3392).
    2 x per_ctx batches of the same reads, everything inside the clock: reads up, the five stages, ranking on the
    device (-b 3), ranked hits down (+ the CSR when a read is handed back), the profile down once per context at the end."""
    import threading
    import numpy as np
    from kasa_amd import capi
    other = capi.Context(dix, k_high, k_low, 3)
    try:
        nnz = csr_buf[1].shape[0]
        bufs = [(csr_buf, rank_buf),
                ((capi.pinned_empty(reads.n + 1, np.uint64), capi.pinned_empty(nnz, np.uint32), capi.pinned_empty(nnz, np.float32)),
                 (capi.pinned_empty(reads.n * 4, np.uint32), capi.pinned_empty(reads.n * 8, capi.RANK_ENTRY)))]
        ctxs = [ctx, other]
        errors, got = [], [0, 0]

        def batch(i):
            c = ctxs[i]
            c.upload(bases, offsets)
            c.encode()
            c.sort_and_range()
            c.lookup_score(True, False)
            meta, ent, flagged = c.rank(den, rclass, 0.0, 3, out=bufs[i][1])
            if flagged:
                c.scores(out=bufs[i][0])
            got[i] = int(ent.shape[0])

        batch(1)                                                               # the second context sizes its buffers outside the clock, as the first has
        start = threading.Barrier(3)

        def work(i):
            try:
                start.wait()
                for _ in range(per_ctx):
                    batch(i)
                ctxs[i].profile()
            except Exception as ex:                                            # pragma: no cover
                errors.append(str(ex)[:200])

        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        if errors:
            return {"pcie_pipelined_error": errors[0]}
        return {"pcie_pipelined_reads_per_s": 2 * per_ctx * reads.n / dt, "pcie_pipelined_s_per_batch": dt / (2 * per_ctx),
                "pcie_pipelined_batches": 2 * per_ctx, "pcie_pipelined_contexts": 2, "pcie_pipelined_ranked_entries_per_batch": got[0]}
    finally:
        other.close()


def pcie_inclusive(ctx, reads, want, ix, k_high, dix=None, k_low=7):
    """Two more passes with the PCIe legs inside the clock (never `value`): host reads in, and out either what the per-read
    file can print (ranked on the device, kasa_batch_rank; page-locked buffers) or the whole CSR (pageable memory, the
    round-1 path).  The file-to-file rate of the C++ driver is measured by file_to_file() below."""
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
        if dix is not None:
            try:
                out.update(pcie_pipelined(ctx, dix, reads, bases, offsets, den, rclass, csr_buf, rank_buf, k_high, k_low))
            except Exception as ex:                                        # (e.g. no memory for a second context)
                out["pcie_pipelined_error"] = str(ex)[:300]
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
                "note": "pcie_inclusive: ONE batch, nothing overlapped: page-locked reads up, device, ranking on the device (-b 3), ranked hits + profile down "
                        "(+ the CSR into page-locked memory when the device hands reads back); "
                        "pcie_pipelined: the same legs for 6 batches through two contexts on two host threads (copies of one batch beside the kernels of another); "
                        "csr_download: the same with the whole CSR down into pageable memory; file_to_file_*: the C++ driver as a child process (text written on the device)"})
    if not want:
        out["pcie_inclusive_reads_per_s"] = out["csr_download_reads_per_s"]
        out["pcie_inclusive_s_per_batch"] = out["csr_download_s_per_batch"]
    return out


def file_to_file(args, ix, reads, device, memories=None):
    """The C++ driver (kASA's `identify` command line over the C ABI) as a child process: FASTQ of the same reads and the
    index files in /dev/shm, JSONL + profile out.  Rate = reads / the driver's own "Time file" (everything but loading the
    index, as the reference reports it, Compare.hpp:3689-3690).  The parent must have released its device memory."""
    import shutil
    import tempfile
    from kasa_amd import build, formats
    exe = build.build_host()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="kasa_bench_", dir=base)
    try:
        t0 = time.perf_counter()
        formats.write_index(ix, os.path.join(d, "idx"), os.path.join(d, "content.txt"))
        L = args.read_len
        bases = reads.bases.reshape(reads.n, L)
        fq = os.path.join(d, "reads.fastq")
        rec = np.empty((reads.n, 2 * L + 16), dtype=np.uint8)          # "@" + 9 digits + "\n" + bases + "\n+\n" + quality + "\n"
        rec[:, 0] = ord("@")
        ids = np.arange(reads.n, dtype=np.int64)
        for c in range(9):
            rec[:, 9 - c] = (ord("0") + (ids // 10 ** c) % 10).astype(np.uint8)
        rec[:, 10] = 10
        rec[:, 11:11 + L] = bases
        rec[:, 11 + L:14 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 14 + L:14 + 2 * L] = ord("I")
        rec[:, 14 + 2 * L] = 10
        rec = rec[:, :15 + 2 * L]
        with open(fq, "wb") as f:
            f.write(np.ascontiguousarray(rec).tobytes())
        del rec
        t_files = time.perf_counter() - t0
        def one(mem):
            cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", fq,
                   "-q", os.path.join(d, "out.jsonl"), "-p", os.path.join(d, "prof.csv"), "--jsonl", "-v", "-m", str(mem),
                   "--device", str(device)]
            for name in ("out.jsonl", "prof.csv"):                  # (truncating 5.5 GB of tmpfs pages inside the child's clock costs 0.4 s)
                try:
                    os.unlink(os.path.join(d, name))
                except OSError:
                    pass
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200,
                               env=dict(os.environ, KASA_ALLOC_TIMING="1"))
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stdout[-400:]}
            t = {}
            slow_alloc = 0.0                                       # hipMalloc calls of more than 20 ms (the library reports them)
            for line in r.stdout.splitlines():
                for key in ("Time fastq", "Time compare", "Time output", "Time file"):
                    if line.startswith("OUT: " + key + ":"):
                        t[key] = float(line.split(":")[2].split()[0])
                if line.startswith("kasa: hipMalloc("):
                    slow_alloc += float(line.split(" took ")[1].split()[0])
            n_batches = sum(1 for line in r.stdout.splitlines() if line.startswith("OUT: Batch of "))
            return {"file_to_file_reads_per_s": reads.n / t["Time file"] if t.get("Time file") else None,
                    "file_to_file_s": t.get("Time file"), "parse_s": t.get("Time fastq"), "device_s": t.get("Time compare"),
                    "text_s": t.get("Time output"), "slow_hipmalloc_s": round(slow_alloc, 3), "child_wall_s_incl_index_load": wall, "batches": n_batches, "memory_gib": mem,
                    "input_bytes": os.path.getsize(fq), "output_bytes": os.path.getsize(os.path.join(d, "out.jsonl"))}
        # the pipelined run (-m small enough for several batches: parse, device and text overlap) is the headline of this leg;
        # the one-batch run (-m large: what the reference does when everything fits its budget) is reported beside it
        mems = memories or [args.f2f_memory, args.f2f_memory_one_batch]
        # (hipMalloc on this platform takes 0-90 ms per GB depending on what other processes have just freed -- the driver
        # clears released VRAM in the background -- so each child starts after the device has been left alone for a while,
        # and what the library saw of it is reported as slow_hipmalloc_s: part of device_s and of the file time)
        time.sleep(args.f2f_settle)
        out = one(mems[0])
        if len(mems) > 1 and "error" not in out:
            time.sleep(args.f2f_settle)
            out["one_batch"] = one(mems[1])
        out["command"] = ("kasa_identify identify --jsonl -m <GiB> -v (FASTQ and index in %s; inputs written in %.1f s; rate = reads / the "
                          "driver's own 'Time file', index load excluded)" % (base or "tmp", t_files))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def long_reads_leg(args, capi, synth, torch, device, read_len=10_000):
    """`quaternary`: long reads against the headline's index.  (a) n_reads x read_len bp reads (a long-read data set: about a third
    of 10 kb reads repeat one of their own 7-letter prefixes, break the fast kernels' order rule and take the general kernel, one
    wavefront per read); (b) ONE 9.6 Mbp contig -- 32 different genomes behind each other -- and (c) one genome 32 times over (every
    group of the read hit 32 times: its longest float chain has 5e7 addends): reads of 16384 k-mers and more are replayed from
    sorted events (kasa_replay.h; round 5: one wavefront, 0.26-0.35 M k-mers/s).  Bases and offsets resident in HBM, the step as
    in the headline: upload_device (geometry) + encode + sort + lookup + group + score."""
    import numpy as np
    g = synth.genomes(args.taxa, args.genome_len, seed=11)
    wide = bool(getattr(args, "wide", False))                             # (--long-reads --wide: the same against the 128-bit index, -k 25 7)
    ix = synth.index_from_genomes(g, device=device, K=25 if wide else 12)
    dix = capi.DeviceIndex(ix, device, check_trie=False)
    ctx = capi.Context(dix, 25 if wide else 12, 7, 3)
    if args.debug_flags:
        ctx.debug_flags(args.debug_flags)
    out = {"index_records": int(ix.n), "index": "128-bit, -k 25 7" if wide else "64-bit, -k 12 7", "step": "kasa_batch_upload_device + encode + sort + lookup + group + score, inputs resident in HBM"}

    def run(name, bases, offsets, steps, what):
        n = int(offsets.shape[0] - 1)
        db, do = capi.DeviceBuffer(bases.shape[0], device), capi.DeviceBuffer(offsets.shape[0] * 8, device)
        db.write(bases)
        do.write(np.ascontiguousarray(offsets, dtype=np.int64))

        def step():
            ctx.profile_reset()
            ctx.upload_resident(db.ptr, do.ptr, n)
            ctx.encode()
            ctx.sort_and_range()
            ctx.lookup_score(True, False)
        step(); step()
        ctx.synchronize()
        ctx.stage_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kmers = ctx.n_kmers
        ca, _, _ = ctx.profile()
        st = ctx.batch_stats()
        km = {k: v[0] / steps for k, v in ctx.kernel_ms().items() if v[1] and k in ("score_general_kernels", "score_replay_kernels", "score_dense_kernel", "score_main_kernel", "group_kernel")}
        out[name] = {"workload": what, "reads": n, "kmers": int(kmers), "ms_per_step": dt * 1e3, "reads_per_s": n / dt, "kmers_per_s": kmers / dt,
                     "identified_fraction": float(ca[-1].sum()) / max(1, kmers),
                     "stage_ms_per_step": {k: v[0] / steps for k, v in ctx.stage_ms().items()}, "kernels_ms_per_step": km,
                     "general_reads": st["general_reads"], "dense_reads": st["dense_reads"], "replay_reads": st["replay_reads"], "replay_events": st["replay_events"]}
        db.close(); do.close()

    n_reads = args.long_reads_n
    read_len = min(read_len, args.genome_len)
    r = synth.reads_from_genomes(g, n_reads, read_len, seed=4242, chunk=4096)
    run("reads_10kb", r.bases, r.offsets, 2, f"{n_reads} synthetic {read_len} bp reads (1 % substitutions) in one batch")
    del r
    rng = np.random.default_rng(3)
    for name, picks, what in (("contig_different_genomes", list(range(0, 64, 2)), "ONE sequence of 9.6 Mbp: 32 different genomes behind each other, 1 % substitutions"),
                              ("contig_one_genome_32_times", [0] * 32, "ONE sequence of 9.6 Mbp: one genome 32 times over, 1 % substitutions")):
        seq = np.concatenate([g[t % args.taxa] for t in picks]).copy()
        m = rng.random(seq.shape[0]) < 0.01
        seq[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(m.sum()))]
        run(name, seq, np.array([0, seq.shape[0]], dtype=np.int64), 2, what)
    ctx.close()
    dix.close()
    return out


def make_comm(rank, world, share, torch, dist, kdist):
    """The C ABI's own RCCL communicator for the profile reduce (N > 1) -> (comm, ranks RCCL reports, how the reduce runs)."""
    comm, rccl_ranks, reduce_how = 0, None, None
    if world > 1:                                        # every rank says which runtimes it runs on (a bad curve starts with a mismatch somewhere)
        try:
            from kasa_amd import capi
            ri = capi.runtime_info()
            log(f"[rank {rank}] hip_runtime {ri.get('hip_runtime')} (built with {ri.get('hip_built')}), rccl_runtime {ri.get('rccl_runtime')}, from {ri.get('runtime_from')}")
        except Exception as ex:
            log(f"[rank {rank}] runtime info: {ex}")
    if world > 1 and not share:
        try:
            comm, rccl_ranks = kdist.rccl_communicator(rank, world)
            reduce_how = "kasa_profile_allreduce (C ABI: limbs packed on the device, ncclAllReduce on the context's stream)"
        except Exception as ex:                          # never lose the measurement to the plumbing: torch.distributed carries the same sum
            comm, reduce_how = 0, "torch.distributed all_reduce of the limbs (C-ABI communicator failed: %s)" % str(ex)[:200]
        ok = torch.tensor([1 if comm else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)        # all ranks the same way
        if int(ok.item()) == 0 and comm:
            kdist.rccl_destroy(comm)
            comm, reduce_how = 0, "torch.distributed all_reduce of the limbs (C-ABI communicator failed on another rank)"
    elif world > 1:
        reduce_how = "gloo all_reduce of the limbs through the host (KASA_BENCH_SHARE_GPU test hook)"
    return comm, rccl_ranks, reduce_how


def bench_partitioned(args, rank, local_rank, world, share, torch, dist, capi, synth, kdist):
    """BASELINE.json configs[4] (C5): an index too large for one GPU, range-partitioned over the ranks at 30-bit prefix
    boundaries (kasa_amd/partition.py, dist.py).  Every rank's slice = the genomes' records of its prefix range plus random
    filler records (the index is synthesised directly, sorted (kmer, taxid) + unique), made on the device;
    every rank also owns --reads reads.  A step = one batch per rank through the exchange: sorted k-mers out (8 B per
    query), event records + taxon lists back (32 B per query + pool), three RCCL all_to_all on device tensors, then the
    profile reduce.  Batches are sized from the HBM that is free once the slice is loaded."""
    from kasa_amd import partition
    k_high, k_low, L = 12, 7, args.read_len
    dev = torch.device("cuda", local_rank)
    t0 = time.perf_counter()
    g = synth.genomes(args.taxa, args.genome_len, seed=11)
    ix = synth.index_from_genomes(g, device=local_rank)
    parts, cuts = partition.split_index(ix, world)
    mine = parts[rank]
    lo = int(cuts[rank]) << 30
    hi = (int(cuts[rank + 1]) if rank + 1 < world else (1 << 30)) << 30
    n_fill = max(0, int(args.part_records) - int(mine.n))
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242 + rank)
    km = torch.empty(mine.n + n_fill, dtype=torch.int64, device=dev)
    td = torch.empty(mine.n + n_fill, dtype=torch.int32, device=dev)
    km[:mine.n] = torch.from_numpy(mine.kmer.astype(np.int64)).to(dev)
    td[:mine.n] = torch.from_numpy(mine.taxid.astype(np.int32)).to(dev)
    step = 1 << 28
    for a in range(mine.n, mine.n + n_fill, step):                       # random 60-bit k-mers of this rank's prefix range, random taxa
        b = min(mine.n + n_fill, a + step)
        km[a:b] = torch.randint(lo, hi, (b - a,), dtype=torch.int64, device=dev, generator=gen)
        td[a:b] = torch.randint(100, 100 + args.taxa, (b - a,), dtype=torch.int32, device=dev, generator=gen)
    td, order = torch.sort(td, stable=True)                                # (kmer, taxid) order: by taxid first, then stably by k-mer
    km = km[order]
    del order
    km, order = torch.sort(km, stable=True)
    td = td[order]
    del order
    keep = torch.ones(km.shape[0], dtype=torch.bool, device=dev)
    keep[1:] = (km[1:] != km[:-1]) | (td[1:] != td[:-1])
    km, td = km[keep], td[keep]
    del keep
    n_rec = int(km.shape[0])
    rec = torch.empty((n_rec, 12), dtype=torch.uint8, device=dev)          # the index file's records: {u64 kmer, u32 taxid}
    rec[:, :8] = km.view(torch.uint8).view(n_rec, 8)
    rec[:, 8:] = td.view(torch.uint8).view(n_rec, 4)
    del km, td
    torch.cuda.synchronize()
    dix = capi.DeviceIndex.from_device_records(rec.data_ptr(), n_rec, 12, ix.content.taxids, local_rank)
    del rec
    torch.cuda.empty_cache()
    log(f"[rank {rank}] slice: {n_rec} records ({n_rec * 12 / 1e9:.1f} GB as a file, {dix.device_bytes / 1e9:.1f} GB in HBM), "
        f"prefixes [{int(cuts[rank])}, {hi >> 30}), {time.perf_counter() - t0:.1f} s")
    owner = capi.Context(dix, k_high, k_low, 3)
    worker = partition.Worker(dix, k_high, k_low, 3)
    free, total = capi.device_memory(local_rank)
    per_q = capi.bytes_per_query(owner) + 64 + 96                          # owner + worker (keys, depth, rep, records, pool) + exchange tensors
    q = max(1, L - 3 * k_low + 1)
    n_reads = int(min(args.reads, max(1000, 0.8 * free / per_q / q)))
    if world > 1:                                                          # everybody the same batch (collective rounds)
        t = torch.tensor([n_reads], dtype=torch.int64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        n_reads = int(t.item())
    reads = synth.reads_from_genomes(g, n_reads, L, seed=1000 + rank)
    stats = {}
    comm, rccl_ranks, reduce_how = make_comm(rank, world, share, torch, dist, kdist)
    acc = {"reduce_s": 0.0}

    def step():
        owner.profile_reset()                                              # a step is a whole "file"
        kdist.partitioned_batch(owner, worker, cuts, 12, reads, not args.profile_only, False, stats=stats)
        if world > 1:
            owner.synchronize()
            t0 = time.perf_counter()
            if comm:
                owner.profile_allreduce(comm)                              # the C ABI's reduce, as in the read-sharded run
            else:
                owner.profile_set_limbs(kdist.allreduce_limbs(owner.profile_limbs(), device=None if share else "cuda"))
            owner.synchronize()
            acc["reduce_s"] += time.perf_counter() - t0

    def fence():
        owner.synchronize(); worker.ctx.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    acc["reduce_s"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    owner.synchronize(); worker.ctx.synchronize()
    mine = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    tdev = "cpu" if share else dev
    tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    every = [torch.zeros(1, dtype=torch.float64, device=tdev) for _ in range(world)]
    dist.all_gather(every, torch.tensor([mine], dtype=torch.float64, device=tdev))
    rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
    ca, cu, _ = owner.profile()
    n_kmers = owner.n_kmers
    out = {"metric": "reads/s in identify (150bp reads vs range-partitioned k=12 index, cross-rank lookup)",
           "value": n_reads * world * args.steps / dt, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
           "data": "synthetic",
           "config": {"workload": f"{n_reads} synthetic {L} bp reads per GPU vs an index of {world} slices, {n_rec} records "
                                  f"({n_rec * 12 / 1e9:.1f} GB) on this rank, -k 12 7, 3 frames, reads uploaded inside the step",
                      "reads_per_gpu": n_reads, "kmers_per_gpu": n_kmers, "slice_records": n_rec, "slice_hbm_bytes": dix.device_bytes,
                      "free_hbm_after_slice_bytes": free, "bytes_per_query_budgeted": per_q,
                      "parallelism": f"index range-partitioned x{world}, reads sharded x{world}",
                      "exchange": "gloo, host-staged (KASA_BENCH_SHARE_GPU test hook)" if share else "RCCL all_to_all on device tensors"},
           "kmers_per_s": n_kmers * world * args.steps / dt,
           "identified_fraction": float(ca[-1].sum()) / max(1, n_kmers * world),
           "rank_step_ms": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms},
           "reduce_ms_per_step": acc["reduce_s"] / max(1, args.steps) * 1e3,
           "exchange_bytes_per_step_this_rank": stats}
    out["config"].update({"reduce": reduce_how, "rccl_ranks": rccl_ranks})
    if comm:
        kdist.rccl_destroy(comm)
    owner.close(); worker.close(); dix.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads of the one batch at N = 1; the largest batch at N > 1")
    ap.add_argument("--total-reads", type=int, default=None, help="reads of all ranks together, taken in batches of at most --reads (\"strong\"); default: --reads on every rank (\"weak\")")
    ap.add_argument("--c4-reads", type=int, default=100_000_000, help="N > 1: total reads of the `c4` leg (BASELINE.json configs[3])")
    ap.add_argument("--no-c4", action="store_true", help="N > 1: skip the `c4` leg")
    ap.add_argument("--no-c2-strong", action="store_true", help="N > 1: skip the `c2_strong` leg (--reads reads in all over the N ranks)")
    ap.add_argument("--no-tertiary", action="store_true", help="skip the crowded-index workload")
    ap.add_argument("--no-quaternary", action="store_true", help="skip the long-read workloads (100 000 x 10 kb reads; one 9.6 Mbp contig)")
    ap.add_argument("--long-reads", action="store_true", help="the long-read workloads as the only measurement")
    ap.add_argument("--long-reads-n", type=int, default=100_000, help="reads of the 10 kb workload")
    ap.add_argument("--crowded", action="store_true", help="the crowded-index workload as the only measurement (profiling)")
    ap.add_argument("--crowded-reads", type=int, default=2_000_000, help="reads of the crowded-index batch (its (event, taxon) contributions per read are ten times the headline's)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes for roofline.traffic")
    ap.add_argument("--pmc-secondary", action="store_true", help="also measure the 128-bit leg's traffic live (two more child passes)")
    ap.add_argument("--taxa", type=int, default=1400)
    ap.add_argument("--genome-len", type=int, default=300_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=300_000, help="reads of the one-thread CPU run")
    ap.add_argument("--cpu-sample-parallel", type=int, default=5_000_000, help="reads of the all-cores CPU run (less when the host's memory is short)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive extra pass and the file-to-file run")
    ap.add_argument("--no-f2f", action="store_true", help="skip the file-to-file run of the C++ driver")
    ap.add_argument("--f2f-memory", type=int, default=12, help="-m of the file-to-file run (GiB; the reference's batch budget): several batches, pipelined")
    ap.add_argument("--f2f-memory-one-batch", type=int, default=1024, help="-m of the second file-to-file run: everything in one batch")
    ap.add_argument("--profile-only", action="store_true", help="no per-read scores (kASA without -q)")
    ap.add_argument("--wide", action="store_true",
                    help="BASELINE.json configs[2] as the only measurement: 128-bit index, -k 25 7")
    ap.add_argument("--secondary", action="store_true", help="(the default at N = 1; kept for older command lines)")
    ap.add_argument("--f2f-settle", type=float, default=8.0, help="seconds the device is left alone before each file-to-file child (VRAM released by the previous process is cleared in the background)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="do not run configs[2] (128-bit index, -k 25 7, same reads) after the headline measurement")
    ap.add_argument("--debug-flags", type=int, default=0, help="kasa_ctx_debug bits for A/B runs of one kernel choice against another (0 = the product path)")
    ap.add_argument("--partitioned", action="store_true", help="BASELINE.json configs[4]: range-partitioned index (see bench_partitioned)")
    ap.add_argument("--part-records", type=float, default=3.0e9, help="--partitioned: index records per rank (36 GB at 3e9)")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)   # (set by run_with_retry)
    args = ap.parse_args()

    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.child and not args.no_pmc and not args.crowded and not args.long_reads and not args.wide
            and not args.partitioned and os.environ.get("KASA_BENCH_NO_RETRY") != "1"):
        raise SystemExit(run_with_retry())             # nothing has touched a GPU in this process
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

    # One GPU: no torch in the process -- the reads lie in plain device buffers of the C ABI (capi.DeviceBuffer) and the library
    # runs on the HIP runtime it was built for.  N > 1 (and --partitioned) needs torch.distributed; its wheel brings its own
    # HIP runtime, which is then the one in the process (capi.share_torch_runtime; the line's `runtime` says which it was).
    torch = None
    DEVICE[0] = local_rank
    if world > 1 or args.partitioned or os.environ.get("KASA_BENCH_TORCH") == "1":
        import torch
    dist = None
    if world > 1 or args.partitioned:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    elif torch is not None:
        torch.cuda.set_device(local_rank)

    from kasa_amd import capi, synth
    from kasa_amd import dist as kdist
    assert capi.device_count() > local_rank, "no HIP device for this rank"

    if args.partitioned:
        out = bench_partitioned(args, rank, local_rank, world, share, torch, dist, capi, synth, kdist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    comm, rccl_ranks, reduce_how = make_comm(rank, world, share, torch, dist, kdist)

    holder = {}

    def one(wide, workload="pairs", total_reads=None, legs=True):
        """One measurement: index (wide: 128-bit, -k 25 7) + the rank's reads resident in HBM + measure().
        workload: "pairs" (sibling genomes, BASELINE's synthetic database) or "crowded" (clades sharing conserved genes);
        total_reads: None = --reads on every rank (weak), else that many over all ranks in batches of at most --reads (strong)."""
        k_high, k_low = (25, 7) if wide else (12, 7)
        rec_bytes = 20 if wide else 12
        t0 = time.perf_counter()
        if workload == "crowded":
            g = synth.genomes_crowded(args.taxa, args.genome_len, seed=11)
        else:
            g = synth.genomes(args.taxa, args.genome_len, seed=11)
        ix = synth.index_from_genomes(g, device=local_rank, K=25 if wide else 12)
        log(f"[rank {rank}] {workload} index: {ix.n} records ({ix.n * rec_bytes / 1e9:.2f} GB on disk layout), "
            f"{ix.trie_prefix.shape[0]} prefixes, {time.perf_counter() - t0:.1f} s")
        t0 = time.perf_counter()
        dix = capi.DeviceIndex(ix, local_rank, check_trie=True)
        ctx = capi.Context(dix, k_high, k_low, 3)
        if args.debug_flags:
            ctx.debug_flags(args.debug_flags)
        n_reads = args.reads if workload != "crowded" else min(args.reads, args.crowded_reads)
        per_rank = n_reads if total_reads is None else max(1, total_reads // world)
        keep, batches, reads = resident_batches(torch, synth, g, per_rank, n_reads, args.read_len, 1000 + rank * 97)
        n_batches = len(batches)
        extra_cfg = {"database": workload}
        if world > 1:
            extra_cfg.update({"total_reads": per_rank * world, "reduce": reduce_how, "rccl_ranks": rccl_ranks,
                              "rccl_ranks_tested": 1})   # (what the builder could run: one GPU per box -- RCCL refuses two ranks on a device; the N-rank path is covered over gloo)
        log(f"[rank {rank}] reads: {per_rank} x {args.read_len} bp in {n_batches} batch(es), {time.perf_counter() - t0:.1f} s")
        warm = args.warmup
        if workload == "crowded":
            args.warmup = max(args.warmup, 2)                      # (its first steps size the key, pool and leftover buffers: 10 GB allocations must not sit in a timed step)
        res = measure(args, ctx, world, dist, share, torch, kdist, batches, comm)
        args.warmup = warm
        out = None
        if rank == 0:
            out = report(args, ctx, reads, ix, world, res, wide, per_rank, n_batches, "weak" if total_reads is None else "strong", extra_cfg)
            if legs and world == 1 and not args.no_e2e and workload == "pairs":
                try:                                               # an extra pass: it must never cost the headline line
                    out["e2e"] = pcie_inclusive(ctx, reads, not args.profile_only, ix, k_high, dix, k_low)
                except Exception as ex:                            # e.g. no memory left for the page-locked buffers
                    out["e2e"] = {"error": str(ex)[:300]}
        ctx.close()
        dix.close()
        for k in keep:
            if hasattr(k, "close"):
                k.close()
        del keep, batches
        if torch is not None:
            torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not wide and legs and workload == "pairs":
            holder["ix"], holder["reads"], holder["k"] = ix, reads, (k_high, k_low)
        if rank == 0 and world == 1 and out["roofline"].get("kernel") and workload == "crowded" and not args.crowded:
            # the crowded workload's dominant kernel: HBM bytes from the counters, measured now (two child passes)
            out["roofline"].update({k: v for k, v in pmc_traffic(args, wide, out["roofline"]["kernel"], not args.no_pmc, crowded=True).items() if k != "insts"})
        if rank == 0 and world == 1 and out["roofline"].get("kernel") and workload == "pairs":
            # HBM bytes of the dominant kernel from the counters (the device is free now)
            live = not args.no_pmc and (args.pmc_secondary if (wide and not args.wide) else True)
            pm = pmc_traffic(args, wide, out["roofline"]["kernel"], live)
            insts = pm.pop("insts", None)
            out["roofline"].update(pm)
            if insts and out["roofline"].get("avg_launch_ms"):
                # instruction issue: a SIMD issues one wavefront instruction per 4 cycles -- (VALU + SALU) x 4 / (1024 SIMDs x 2.4 GHz)
                pred = (insts["valu"] + insts["salu"]) * 4.0 / (1024 * 2.4e9) * 1e3
                out["roofline"]["third_bound"] = {"bound": "issue", "kernel": out["roofline"]["kernel"], "wave_insts_valu": insts["valu"],
                                                  "wave_insts_salu": insts["salu"], "predicted_ms": pred,
                                                  "frac": pred / out["roofline"]["avg_launch_ms"], "source": insts["source"],
                                                  "formula": "(SQ_INSTS_VALU + SQ_INSTS_SALU) * 4 cycles / (1024 SIMDs * 2.4 GHz)"}
        return out

    KEEP = ("metric", "value", "unit", "ms_per_step", "scaling", "dtype", "config", "kmers_per_s", "identified_fraction", "rank_step_ms",
            "reduce_ms_per_step", "upload_ms_per_batch", "batch", "stage_ms_per_step", "roofline", "kernels")
    if args.long_reads:
        out = long_reads_leg(args, capi, synth, torch, local_rank)
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    if args.crowded:
        out = one(False, workload="crowded", legs=False)
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    out = one(args.wide, total_reads=args.total_reads)
    if rank == 0 and world == 1 and not args.wide:
        ix, reads, (k_high, k_low) = holder["ix"], holder["reads"], holder["k"]
        if not args.no_e2e and not args.no_f2f:
            try:                                                   # the device is free now: the C++ driver as a child process
                f2f = file_to_file(args, ix, reads, local_rank)
            except Exception as ex:
                f2f = {"error": str(ex)[:300]}
            out.setdefault("e2e", {}).update(f2f)
        if not args.no_cpu:                                         # the CPU baseline is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline_report(ix, reads, k_high, k_low, args)
        holder.clear()
        del ix, reads
    if world == 1 and not args.wide and not args.no_secondary:
        sec = one(True, legs=False)
        if rank == 0 and out is not None and sec is not None:
            out["secondary"] = {k: sec[k] for k in KEEP}
    if world == 1 and not args.wide and not args.no_tertiary:
        try:
            ter = one(False, workload="crowded", legs=False)
        except Exception as ex:                                     # never lose the headline to the extra workload
            ter = {"error": str(ex)[:400]}
        if rank == 0 and out is not None and ter is not None:
            out["tertiary"] = {k: ter[k] for k in KEEP} if "error" not in ter else ter
    if world == 1 and not args.wide and not args.no_quaternary:
        try:
            qua = long_reads_leg(args, capi, synth, torch, local_rank)
        except Exception as ex:                                     # never lose the headline to the extra workload
            qua = {"error": str(ex)[:400]}
        if rank == 0 and out is not None:
            out["quaternary"] = qua
    if world > 1 and args.total_reads is None and not args.no_c2_strong:
        # BASELINE.json's metric read literally: 10 M reads IN ALL at 1/2/4/8 GPUs ("strong": 10 M / N per rank, one batch each)
        c2s = one(args.wide, total_reads=args.reads, legs=False)
        if rank == 0 and out is not None and c2s is not None:
            out["c2_strong"] = {k: c2s[k] for k in KEEP}
    if world > 1 and args.total_reads is None and not args.no_c4:
        c4 = one(args.wide, total_reads=args.c4_reads, legs=False)   # BASELINE.json configs[3]: 100 M reads in all ("strong")
        if rank == 0 and out is not None and c4 is not None:
            out["c4"] = {k: c4[k] for k in KEEP}
    if rank == 0:
        out["runtime"] = capi.runtime_info()                        # HIP / RCCL the library was built with and runs on (one runtime per process)
        out.setdefault("attempts", 1)
        out.setdefault("retried", False)
        out["summary"] = summary_of(out)                            # LAST key, under 1900 characters: a reader that keeps only the line's tail still sees every leg
        print(json.dumps(out), flush=True)
    if comm:
        kdist.rccl_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def summary_of(out):
    """The line in small, as its last key (the driver's record keeps the standard keys and the last 2000 characters of stdout):
    ms per stage and per kernel of the headline, the end-to-end rates, one figure per further leg."""
    r1 = lambda v: None if v is None else round(float(v), 1)
    def leg(o):
        if not isinstance(o, dict) or "value" not in o:
            return o if o is None else {"error": str(o.get("error"))[:80]} if isinstance(o, dict) else None
        return {"reads_per_s": round(o["value"]), "ms": r1(o["ms_per_step"]), "stage_ms": {k: r1(v) for k, v in o.get("stage_ms_per_step", {}).items() if v},
                "roofline": {"kernel": (o.get("roofline", {}).get("kernel") or "")[:24], "ms": r1(o.get("roofline", {}).get("avg_launch_ms")),
                             "frac": None if o.get("roofline", {}).get("frac") is None else round(o["roofline"]["frac"], 3)}}
    sm = {"stage_ms": {k: r1(v) for k, v in out.get("stage_ms_per_step", {}).items() if v},
          "kernel_ms": {k: r1(v.get("avg_launch_ms", v.get("ms_per_step"))) for k, v in out.get("kernels", {}).items()}}
    if isinstance(out.get("record_placement"), dict):
        sm["record_buffer_g_stores_per_s"] = out["record_placement"].get("candidates_g_records_per_s")   # (candidates timed; the best is kept)
    e = out.get("e2e") or {}
    sm["e2e_reads_per_s"] = {k.replace("_reads_per_s", ""): round(v) for k, v in e.items() if k.endswith("_reads_per_s") and isinstance(v, (int, float))}
    for name in ("secondary", "tertiary", "c2_strong", "c4"):
        if name in out:
            sm[name] = leg(out[name])
    q = out.get("quaternary")
    if isinstance(q, dict):
        sm["quaternary"] = {k: {"kmers_per_s": round(v["kmers_per_s"]), "ms": r1(v["ms_per_step"]), "general_reads": v["general_reads"], "replay_reads": v["replay_reads"]}
                            for k, v in q.items() if isinstance(v, dict) and "kmers_per_s" in v} or {"error": str(q.get("error"))[:80]}
    txt = json.dumps(sm)
    if len(txt) > 1900:                                              # (never longer than the tail a reader keeps)
        sm.pop("kernel_ms", None)
    return sm


def host_cpus():
    """Logical CPUs the machine has, those this process may run on (affinity mask), and the CPU time a cgroup quota allows
    per second of wall time -- a container can see 256 CPUs and be throttled to a handful; `usable` is what the CPU baseline
    starts threads for."""
    out = {"logical": os.cpu_count() or 1}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        out["affinity"] = out["logical"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except Exception:
            continue
    out["cgroup_quota_cpus"] = quota
    usable = min(out["logical"], out["affinity"])
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    out["usable"] = usable
    return out


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
