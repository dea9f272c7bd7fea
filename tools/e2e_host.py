#!/usr/bin/env python3
"""End-to-end timing of the C++ driver (kasa_amd/host/kasa_identify) on files: writes a synthetic index
(bench.py's genomes, scaled by --taxa) and a FASTQ file to a scratch directory, then runs the driver.
Not part of bench.py's metric (which times the device path on resident inputs); this is what a user
of the command line sees, file parsing and text output included."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--taxa", type=int, default=200)
    ap.add_argument("--genome-len", type=int, default=300_000)
    ap.add_argument("--fmt", default="--jsonl")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--memory", type=int, default=0, help="-m <GiB> of the driver (the reference's batch budget; 0 = its default 5)")
    args = ap.parse_args()
    import numpy as np
    from kasa_amd import build, formats, synth
    exe = build.build_host()
    d = tempfile.mkdtemp(prefix="kasa_e2e_")
    g = synth.genomes(args.taxa, args.genome_len, seed=11)
    ix = synth.index_from_genomes(g)
    formats.write_index(ix, os.path.join(d, "idx"), os.path.join(d, "content.txt"))
    rb = synth.reads_from_genomes(g, args.reads, 150, seed=5)
    t0 = time.perf_counter()
    bases = rb.bases.reshape(args.reads, 150)
    with open(os.path.join(d, "reads.fastq"), "wb") as f:
        qual = b"I" * 150
        for a in range(0, args.reads, 100000):
            blk = bases[a:a + 100000]
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (a + i, blk[i].tobytes(), qual) for i in range(blk.shape[0])))
    print("fastq written in %.1f s (%d MB)" % (time.perf_counter() - t0, os.path.getsize(os.path.join(d, "reads.fastq")) >> 20), flush=True)
    cmd = [exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq"),
           "-q", os.path.join(d, "out.txt"), "-p", os.path.join(d, "prof.csv"), args.fmt, "-v"]
    if args.threads:
        cmd += ["-n", str(args.threads)]
    if args.memory:
        cmd += ["-m", str(args.memory)]
    import resource
    t0 = time.perf_counter()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dt = time.perf_counter() - t0
    lines = r.stdout.splitlines()
    print("\n".join(lines[:4] + ["... %d batches ..." % sum(1 for l in lines if l.startswith("OUT: Batch of "))] + [l for l in lines[-8:] if not l.startswith("OUT: Batch of ")]))
    rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / (1 << 20)
    print("driver: %.2f s wall for %d reads = %.0f reads/s; output %d MB; largest resident set of the driver %.1f GB (input file %.1f GB)"
          % (dt, args.reads, args.reads / dt, os.path.getsize(os.path.join(d, "out.txt")) >> 20, rss, os.path.getsize(os.path.join(d, "reads.fastq")) / (1 << 30)))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
