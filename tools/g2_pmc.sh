#!/bin/bash
# counter passes of the group kernels on the A/B probe's workload (one pass per rocprofv3 run, counters only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FL=${1:-0}
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  rocprofv3 --pmc $P --kernel-include-regex "group2?_kernel" --output-format csv -d $R/gpurun_out/g2pmc$i -- python3 $R/tools/ab_probe.py --flags $FL --rank-flags "" --rounds 1 --steps 1 > /dev/null 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/gpurun_out/g2pmc*/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name=re.sub(r"\(.*","",row["Kernel_Name"]).replace("void ","")
        per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name,cs in per.items():
        for c,v in cs.items(): acc[name][c]=max(v)
for name,cs in acc.items():
    print(name)
    for c,v in sorted(cs.items()): print("   %-24s %.4g"%(c,v))
    if "SQ_INSTS_VALU" in cs: print("   issue ms at 4 cycles: %.1f"%((cs["SQ_INSTS_VALU"]+cs["SQ_INSTS_SALU"])*4/(1024*2.4e9)*1e3))
PY
