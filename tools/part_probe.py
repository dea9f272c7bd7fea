"""kasa_identify over the bench's synthetic index (4.2e8 records) as ONE index object and as range partitions of at most
--part-records records (KASA_INDEX_PART_RECORDS): same bytes out, and what the partitions cost.  Files in /dev/shm.
    python tools/part_probe.py [--reads 2000000] [--part-records 100000000]
Prints one JSON object."""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--part-records", type=int, default=100_000_000)
    ap.add_argument("--memory", type=int, default=1024)
    args = ap.parse_args()
    import numpy as np
    from kasa_amd import build, formats, synth
    g = synth.genomes(1400, 300_000, seed=11)
    ix = synth.index_from_genomes(g, device=0, K=12)
    reads = synth.reads_from_genomes(g, args.reads, 150, seed=1000)
    exe = build.build_host()
    d = tempfile.mkdtemp(prefix="kasa_part_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    try:
        formats.write_index(ix, os.path.join(d, "idx"), os.path.join(d, "content.txt"))
        fq = os.path.join(d, "reads.fastq")
        with open(fq, "wb") as f:
            bases = reads.bases.reshape(reads.n, 150)
            for a in range(0, reads.n, 200000):
                blk = bases[a:a + 200000]
                f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (a + i, blk[i].tobytes(), b"I" * 150) for i in range(blk.shape[0])))
        res = {"reads": reads.n, "index_records": int(ix.n)}
        sums = {}
        for name, env in (("one_index", {}), ("partitions", {"KASA_INDEX_PART_RECORDS": str(args.part_records)})):
            out, prof = os.path.join(d, "out_%s.jsonl" % name), os.path.join(d, "prof_%s.csv" % name)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", fq, "-q", out, "-p", prof,
                                "--jsonl", "-v", "-m", str(args.memory)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200,
                               env=dict(os.environ, KASA_HOST_TIMING="1", **env))
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                res[name] = {"error": r.stdout[-600:]}
                continue
            t = {}
            for line in r.stdout.splitlines():
                for key in ("Time compare", "Time file"):
                    if line.startswith("OUT: " + key + ":"):
                        t[key] = float(line.split(":")[2].split()[0])
                if line.startswith("OUT: Index of"):
                    t["partitions"] = int(line.split()[6])
                if line.startswith("OUT: device stages"):
                    t["device_stages_ms"] = line.split(":", 2)[2].strip()
            sums[name] = [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in (out, prof)]
            res[name] = {"file_s": t.get("Time file"), "device_s": t.get("Time compare"), "partitions": t.get("partitions", 1), "wall_s_incl_index_load": round(wall, 2),
                         "device_stages_ms": t.get("device_stages_ms")}
            os.unlink(out)
        res["same_bytes"] = len(sums) == 2 and sums["one_index"] == sums["partitions"]
        print(json.dumps(res))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
