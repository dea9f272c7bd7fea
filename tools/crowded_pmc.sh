#!/bin/bash
# counter passes of the crowded-index workload's kernels (bench.py --crowded), one pass per rocprofv3 run, counters only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/crowded_pmc.json}
KR="score_dense_kernel|group2?_kernel|score_kernel|profile_group_table_kernel|score_main_kernel|score_other_flat_kernel|row_merge_bitmap_kernel|profile_reduce_kernel|pass_kernel"
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $P --kernel-include-regex "$KR" --output-format csv -d $R/gpurun_out/crpmc$i -- python3 $R/bench.py --crowded --no-pmc --steps 1 --warmup 1 > /dev/null 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections,re,json
acc=collections.defaultdict(dict)
for f in glob.glob("$R/gpurun_out/crpmc*/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name=re.sub(r"\(.*","",row["Kernel_Name"]).replace("void ","")
        per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name,cs in per.items():
        for c,v in cs.items(): acc[name][c]=max(v)
for name,cs in acc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs: cs["hbm_bytes_per_launch"]=cs["FETCH_SIZE"]*2048+cs["WRITE_SIZE"]*1024
    if "SQ_INSTS_VALU" in cs: cs["issue_ms_at_4_cycles"]=(cs["SQ_INSTS_VALU"]+cs["SQ_INSTS_SALU"])*4/(1024*2.4e9)*1e3
    print(name, {k:(round(v,3) if v<1e6 else float("%.4g"%v)) for k,v in cs.items()})
json.dump({"workload":"bench.py --crowded --steps 1 --warmup 1 (2 M reads, crowded index): per kernel the largest dispatch","kernels":acc}, open("$OUT","w"), indent=1)
PY
