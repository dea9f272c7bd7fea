#!/bin/bash
# per-kernel times of one A/B probe run (rocprofv3 --kernel-trace --stats) + why group2_kernel lists tiles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KASA_DEBUG_WHY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g2prof -- python3 $R/tools/ab_probe.py --flags ${1:-0} --rank-flags "" --rounds 1 --steps 2 > $R/gpurun_out/g2_probe.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/g2prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:18]:
        print(r["Name"][:70], r["Calls"], round(float(r["TotalDurationNs"])/1e6,1), round(float(r["AverageNs"])/1e6,3))
PY
grep -E "group2 left|flags" $R/gpurun_out/g2_probe.log | tail -8
