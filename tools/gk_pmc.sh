#!/bin/bash
# counter passes of the general score kernel (and the other score kernels) on the long-read workload: bench.py --long-reads
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-20000}
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  rocprofv3 --pmc $P --kernel-include-regex "score_kernel|score_dense_kernel|flush_positions_kernel|score_main_kernel" --output-format csv -d $R/gpurun_out/gkpmc$i -- python3 $R/bench.py --long-reads --long-reads-n $N > /dev/null 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/gpurun_out/gkpmc*/**/*counter_collection.csv", recursive=True):
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
