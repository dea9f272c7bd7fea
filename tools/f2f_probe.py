#!/usr/bin/env python3
"""File-to-file timing of the C++ driver with the host-side breakdown (KASA_HOST_TIMING=1): python tools/f2f_probe.py [reads] [-m GiB ...]"""
import os, subprocess, sys, tempfile, shutil, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kasa_amd import build, formats, synth
import bench

class A: pass
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
mems = [int(x) for x in sys.argv[2:]] or [1024, 12]
g = synth.genomes(1400, 300_000, seed=11)
ix = synth.index_from_genomes(g)
reads = synth.reads_from_genomes(g, n, 150, seed=1000)
os.environ["KASA_HOST_TIMING"] = "1"
os.environ["KASA_ALLOC_TIMING"] = os.environ.get("F2F_ALLOC_MS", "20")
for m in mems:
    a = A(); a.read_len = 150; a.f2f_memory = m
    # bench.file_to_file prints nothing of the child's output: run the same command here to see the host timing line
    d = tempfile.mkdtemp(prefix="kasa_f2f_", dir="/dev/shm")
    try:
        formats.write_index(ix, os.path.join(d, "idx"), os.path.join(d, "content.txt"))
        L = 150
        rec = np.empty((reads.n, 2 * L + 15), dtype=np.uint8)
        rec[:, 0] = ord("@")
        ids = np.arange(reads.n, dtype=np.int64)
        for c in range(9):
            rec[:, 9 - c] = (ord("0") + (ids // 10 ** c) % 10).astype(np.uint8)
        rec[:, 10] = 10
        rec[:, 11:11 + L] = reads.bases.reshape(reads.n, L)
        rec[:, 11 + L:14 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 14 + L:14 + 2 * L] = ord("I")
        rec[:, 14 + 2 * L] = 10
        open(os.path.join(d, "reads.fastq"), "wb").write(rec.tobytes())
        del rec
        nlist = os.environ.get("F2F_THREADS", "")
        for threads in ([["-n", x] for x in nlist.split(",")] if nlist else ([] if m != mems[0] else [["-n", "16"]]) + [[]]):
            cmd = [build.build_host(), "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", os.path.join(d, "reads.fastq"),
                   "-q", os.path.join(d, "out.jsonl"), "-p", os.path.join(d, "prof.csv"), "--jsonl", "-v", "-m", str(m)] + threads
            for name in ("out.jsonl", "prof.csv"):
                try: os.unlink(os.path.join(d, name))
                except OSError: pass
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            print("== -m", m, " ".join(threads), "wall %.2f s" % (time.perf_counter() - t0))
            print("\n".join(l for l in r.stdout.splitlines() if l.startswith("OUT: Time") or "host timing" in l or "device stages" in l or l.startswith("ERROR") or (l.startswith("kasa:") and not os.environ.get("F2F_QUIET"))), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
