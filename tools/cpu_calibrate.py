#!/usr/bin/env python3
"""Calibrate bench.py's CPU baseline (oracle/, a restatement) against the REAL reference, in the development container
where the reference's shipped binary runs (binaries/kASA_linux v1.4.9).

Same reduced-scale synthetic input for both: 20 taxa x 100 kb (odd taxa 3 % off their predecessor), 200 000 reads x 150 bp
with 1 % errors, -k 12 7, three frames, per-read JSONL + profile.  Times `kASA identify -r -n 1 / -n <cores>` (whole
process minus index load is not separable there, so wall time of the run) and the oracle's threaded batch on the parsed
reads, and writes profiles/cpu_calibration.json: reads/s of both and their ratio.

    python tools/cpu_calibrate.py            # needs /root/reference
"""
import json
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_fixtures as mf   # noqa: E402  (generator helpers + how to start the reference binary)
from kasa_amd import formats, reads as rd   # noqa: E402
from oracle import oracle   # noqa: E402


def main():
    cores = os.cpu_count() or 1
    G, L, NR = 20, 100_000, 200_000
    out = tempfile.mkdtemp(prefix="kasa_cal_")
    rng = random.Random(11)
    genomes = []
    for g in range(G):
        genomes.append(mf.mutate(genomes[g - 1], 0.03, rng) if g % 2 else "".join(rng.choice("ACGT") for _ in range(L)))
    mf.write_db(out, genomes)
    rng = random.Random(12)
    with open(os.path.join(out, "reads.fastq"), "w") as f:
        for r in range(NR):
            g = rng.randrange(G)
            p = rng.randrange(L - 150)
            f.write("@read%d_t%d\n%s\n+\n%s\n" % (r, g, mf.mutate(genomes[g][p:p + 150], 0.01, rng), "I" * 150))
    mf.run(["build", "-c", "content.txt", "-d", "idx", "-i", "db.fasta", "-m", "8", "-n", str(cores)], out)
    res = {"input": f"{G} taxa x {L} bp, {NR} reads x 150 bp, 1 % errors, -k 12 7, 3 frames", "cores": cores,
           "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown"),
           "reference_binary": "binaries/kASA_linux v1.4.9 (AVX build)"}
    for n in (1, cores):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            mf.run(["identify", "-c", "content.txt", "-d", "idx", "-i", "reads.fastq", "-q", "out.jsonl", "-p", "prof.csv", "--jsonl",
                    "-r", "-m", "16", "-n", str(n)], out)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[f"reference_reads_per_s_n{n}"] = NR / best
        res[f"reference_wall_s_n{n}"] = best
    ix = formats.load_index(os.path.join(out, "idx"), os.path.join(out, "content.txt"))
    batch = rd.parse_reads(os.path.join(out, "reads.fastq"))
    iv = oracle.IndexView(ix)
    p = oracle.params(12, 7, 3)
    for n in (1, cores):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            oracle.identify_threaded(iv, batch.bases, batch.offsets, p, n)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[f"oracle_reads_per_s_n{n}"] = NR / best
        res[f"oracle_wall_s_n{n}"] = best
    res["oracle_over_reference_n1"] = res["oracle_reads_per_s_n1"] / res["reference_reads_per_s_n1"]
    res[f"oracle_over_reference_n{cores}"] = res[f"oracle_reads_per_s_n{cores}"] / res[f"reference_reads_per_s_n{cores}"]
    res["note"] = ("the reference's wall time includes reading the FASTQ, its info pre-pass, ranking and writing 200 000 JSONL lines and "
                   "loading the 24 MB index; the oracle's covers encode + sort + ranges + lookup/score of the parsed reads only")
    with open(os.path.join(ROOT, "profiles", "cpu_calibration.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
