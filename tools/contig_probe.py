"""One long sequence through kasa_identify: how fast the device replays the events of ONE read.  Two inputs over the bench's
synthetic index (1400 genomes x 300 kb): (a) 32 DIFFERENT genomes behind each other (9.6 Mbp, 1 % substitutions: every k-mer
matches, hardly any repeats itself) and (b) 32 copies of ONE genome (every k-mer 32 times in the read: its groups stay open
across the read's later queries, the general kernel's pending window does the bookkeeping).  Files in /dev/shm.
    python tools/contig_probe.py
Prints one JSON object."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from kasa_amd import build, formats, synth
    g = synth.genomes(1400, 300_000, seed=11)
    ix = synth.index_from_genomes(g, device=0, K=12)
    exe = build.build_host()
    d = tempfile.mkdtemp(prefix="kasa_contig_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    try:
        formats.write_index(ix, os.path.join(d, "idx"), os.path.join(d, "content.txt"))
        rng = np.random.default_rng(3)
        res = {"index_records": int(ix.n)}
        only = os.environ.get("KASA_CONTIG_ONLY")
        for name, picks in (("different_genomes", list(range(0, 64, 2))), ("one_genome_32_times", [0] * 32)):
            if only and name != only:
                continue
            seq = np.concatenate([g[t] for t in picks])
            m = rng.random(seq.shape[0]) < 0.01
            seq = seq.copy()
            seq[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(m.sum()))]
            fa = os.path.join(d, name + ".fasta")
            with open(fa, "wb") as f:
                f.write(b">contig\n")
                body = seq.tobytes()
                f.write(b"\n".join(body[i:i + 70] for i in range(0, len(body), 70)) + b"\n")
            out, prof = os.path.join(d, "out.jsonl"), os.path.join(d, "prof.csv")
            r = subprocess.run([exe, "identify", "-c", os.path.join(d, "content.txt"), "-d", os.path.join(d, "idx"), "-i", fa, "-q", out, "-p", prof,
                                "--jsonl", "-v", "-m", "8", "-n", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200,
                               env=dict(os.environ, KASA_HOST_TIMING="1"))
            if r.returncode != 0:
                res[name] = {"error": r.stdout[-500:]}
                continue
            t = {"letters": int(seq.shape[0])}
            for line in r.stdout.splitlines():
                if line.startswith("OUT: Time file:"):
                    t["file_s"] = float(line.split(":")[2].split()[0])
                if line.startswith("OUT: device stages"):
                    t["device_stages_ms"] = line.split(":", 2)[2].strip()
                if line.startswith("OUT: Number of k-mers in input:"):
                    t["kmers"] = int(line.split(":")[2].split()[0])
                    t["identified_percent"] = float(line.split("which")[1].split()[0])
                if line.startswith("kasa-probe:"):                   # (an instrumented build of the library, preloaded)
                    t.setdefault("probe", []).append(line)
                if line.startswith("OUT: Batch of"):
                    t.setdefault("batches", []).append(line[5:])
            if t.get("kmers") and t.get("file_s"):
                t["kmers_per_s"] = t["kmers"] / t["file_s"]
            res[name] = t
        print(json.dumps(res))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
