#!/usr/bin/env python3
"""Where the device idles inside a step: rocprofv3 --kernel-trace of a short bench run, then the gaps between consecutive
dispatches of the LAST step (between the last two encode_kernel launches) -- run on the GPU box:  python3 tools/gap_probe.py"""
import csv, glob, os, subprocess, sys
out = "gpurun_out/gaps"
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", out, "--", "python3", "bench.py", "--steps", "2", "--warmup", "1",
                "--no-cpu", "--no-e2e", "--no-secondary"] + os.environ.get("GAP_ARGS", "").split(), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
rows.sort()
enc = [i for i, r in enumerate(rows) if r[2].startswith("void encode_kernel") or "encode_kernel" in r[2]]
a, b = enc[-2], enc[-1]                       # one whole step between the last two encodes
step = rows[a:b]
busy = sum(e - s for s, e, _ in step)
span = rows[b][0] - rows[a][0]
print("step span %.2f ms, kernels+copies %.2f ms, idle %.2f ms, %d dispatches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(step)))
gaps = []
for i in range(a, b):
    g = rows[i + 1][0] - rows[i][1]
    gaps.append((g, rows[i][2], rows[i + 1][2], (rows[i][1] - rows[a][0]) / 1e6))
gaps.sort(reverse=True)
import collections
acc = collections.OrderedDict()
for s_, e_, n_ in step:
    n_ = n_.replace("void ", "")[:64]
    acc.setdefault(n_, [0, 0]); acc[n_][0] += e_ - s_; acc[n_][1] += 1
for n_, (t_, c_) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:22]:
    print("%8.3f ms x%-3d %s" % (t_ / 1e6, c_, n_))
for g, x, y, t in gaps[:6]:
    print("%8.3f ms idle at t=%7.2f ms after %-50s before %s" % (g / 1e6, t, x, y))
