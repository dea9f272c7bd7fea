#!/usr/bin/env python3
"""Collect the round's profile evidence on the GPU box (run from the repo root; writes into gpurun_out/profiles/):
  * rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-cpu --no-e2e` -> <tag>_kernel_stats_bench_10M.{csv,md}
  * PMC passes (separate runs, counters only) for the hand-written kernels of one 10 M-read step -> <tag>_kernel_pmc.json:
    FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md section HBM) + WRITE_SIZE = HBM bytes per launch, SQ_INSTS_VALU / SALU,
    SQ_WAVES, SQ_BUSY_CYCLES, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY
  * the bench lines (headline, and with --secondary the 128-bit configuration) -> <tag>_bench_1gpu.json
    python3 tools/make_profiles.py r05 [--wide | --crowded] [--pmc-only]
"""
import csv, glob, json, os, re, subprocess, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r06"
WIDE = "--wide" in sys.argv          # the 128-bit configuration (C3): its own kernel summary and counter passes, files <tag>_*_wide.*
CROWDED = "--crowded" in sys.argv    # the crowded-index workload (`tertiary` of the bench line): files <tag>_*_crowded.*
SFX = "_wide" if WIDE else ("_crowded" if CROWDED else "")
WARGS = ["--wide"] if WIDE else (["--crowded", "--warmup", "2"] if CROWDED else [])
out = "gpurun_out/profiles"
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
KERNELS = "esr_emit_kernel|esr_chain_big_kernel|lookup_tile_kernel|group_kernel|group2_kernel|score_dense_kernel|score_kernel|profile_group_accum_kernel|profile_reduce_kernel|bucket_rank64_kernel|score_main_kernel|score_other_flat_kernel|score_other_flat16_kernel|score_other_kernel|row_merge_bitmap_kernel|profile_table_kernel|profile_group_table_kernel|encode_kernel|pass_kernel|hist_kernel|bucket_rank32_kernel|bucket_rank_kernel|row_copy_kernel"
sys.path.insert(0, os.getcwd())
import bench as _bench
SHA = _bench.source_sha16()


def run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, env=env, **kw)


PMC_ONLY = "--pmc-only" in sys.argv
# 1. kernel stats
d = os.path.join(out, "stats" + SFX)
if PMC_ONLY:
    pass
else:
  with open(os.path.join(out, tag + "_bench_under_rocprof" + SFX + ".json"), "w") as f:
      run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--steps", "3", "--warmup", "1",
           "--no-cpu", "--no-e2e", "--no-secondary", "--no-tertiary", "--no-quaternary", "--no-pmc"] + WARGS, stdout=f, stderr=subprocess.DEVNULL)
  src = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
  rows = list(csv.DictReader(open(src)))
  with open(os.path.join(out, tag + "_kernel_stats_bench_10M" + SFX + ".csv"), "w") as f:
      w = csv.writer(f)
      w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
      for r in rows:
          w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
  md = subprocess.run([sys.executable, "tools/kernel_stats.py", d, "40"], stdout=subprocess.PIPE, text=True).stdout
  open(os.path.join(out, tag + "_kernel_stats_bench_10M" + SFX + ".md"), "w").write(
      "Source: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-secondary --no-tertiary --no-quaternary --no-pmc" + "".join(" " + x for x in WARGS) + "` on one MI355X\n" +
      "(" + ("2 M reads x 150 bp, crowded 4.2e8-record 64-bit index (clades of 50-200 taxa sharing conserved genes), -k 12 7" if CROWDED else
             "10 M reads x 150 bp, 4.2e8-record " + ("128-bit index, -k 25 7" if WIDE else "64-bit index, -k 12 7")) + "; kernel sources " + SHA + ").  The run holds the index build (one call of encode / the sort of its own) plus " + ("2 warm-up\n" if CROWDED else "1 warm-up\n") +
      "and 3 timed steps.  The warm-up step launches group / score_main / score_other twice where a first attempt only sizes a buffer (it stops early\n"
      "and is repeated with the capacity it asked for): such a short call lowers the average here below the per-launch average of bench.py's HIP\n"
      "events.  Full names: the .csv next to this file.\n\n" + md)

# 2. PMC passes (counters only; one step of 10 M reads)
passes = ["SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS",
          "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA", "FETCH_SIZE", "WRITE_SIZE"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i, p in enumerate(passes):
    dd = os.path.join(out, "pmc%d%s" % (i, SFX))
    run(["rocprofv3", "--pmc"] + p.split() + ["--kernel-include-regex", KERNELS, "--output-format", "csv", "-d", dd, "--", "python3", "bench.py",
         "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e", "--no-secondary", "--no-tertiary", "--no-quaternary", "--no-pmc"] + WARGS, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in glob.glob(dd + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"[<(].*", "", row["Kernel_Name"]).replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {"source_sha16": SHA,
       "command": "rocprofv3 --pmc <one group per run> --kernel-include-regex '" + KERNELS + "' -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --no-secondary --no-tertiary --no-quaternary --no-pmc" + "".join(" " + x for x in WARGS),
       "workload": ("2M x 150bp reads vs the crowded 4.2e8-record index (bench.py --crowded)" if CROWDED else "10M x 150bp reads vs 419951000-record index (bench.py default)") + "; per kernel the LARGEST dispatch (the 1.3e9-query batch; the index build "
                   "launches encode/lookup once on its own input)",
       "correction": "hbm_bytes_per_launch = FETCH_SIZE[KB] x 1024 x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM; exact for "
                     "16-B-per-lane streaming reads, uncalibrated for narrower ones) + WRITE_SIZE[KB] x 1024"}
for name, cs in acc.items():
    e = {c: max(v) for c, v in cs.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2
        e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    res[{"row_merge_bitmap_kernel": "row_merge_kernel", "score_other_flat_kernel": "score_other_kernel", "score_other_flat16_kernel": "score_other_kernel"}.get(name, name)] = e
json.dump(res, open(os.path.join(out, tag + "_kernel_pmc" + SFX + ".json"), "w"), indent=1)

# 3. the bench line(s).  (Counter passes of this run's kernel sources that are not committed yet -- e.g. the --wide ones made a
# moment ago -- are put where bench.py looks for its fallback, so that the line's `secondary.roofline.traffic` is filled.)
import shutil
for name in (tag + "_kernel_pmc.json", tag + "_kernel_pmc_wide.json", tag + "_kernel_pmc_crowded.json"):
    src = os.path.join(out, name)
    try:
        if os.path.exists(src) and json.load(open(src)).get("source_sha16") == SHA:
            shutil.copy(src, os.path.join("profiles", name))
    except Exception:
        pass
if PMC_ONLY or WIDE or CROWDED:
    for k, e in res.items():
        if isinstance(e, dict) and "SQ_INSTS_VALU" in e:
            ins = e["SQ_INSTS_VALU"] + e["SQ_INSTS_SALU"]
            print("%-24s insts %.2fe9 -> %.1f ms at 4 cycles; HBM %.1f GB" % (k, ins / 1e9, ins * 4 / (1024 * 2.4e9) * 1e3, e.get("hbm_bytes_per_launch", 0) / 1e9))
    sys.exit(0)
args = ["python3", "bench.py", "--steps", "5", "--warmup", "2"]          # (secondary, file to file and CPU baseline are part of the default line)
with open(os.path.join(out, tag + "_bench_1gpu.json"), "w") as f:
    run(args, stdout=f, stderr=subprocess.DEVNULL)
print(open(os.path.join(out, tag + "_bench_1gpu.json")).read()[:600])
