python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5_t7.log 2>&1; tail -3 gpurun_out/r5_t7.log
python tools/ab_probe.py --flags 0,67108864 --rank-flags "" --rounds 1 --steps 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('flags',d['flags'],'group',d['kernel_ms']['group_kernel'],'tables',d['kernel_ms']['profile_table_kernels'],'same',d['profile_same_as_flags0'], d['stage_ms'])"
python bench.py --crowded --no-pmc --steps 2 --warmup 2 > gpurun_out/r5_crowded3.json 2> gpurun_out/r5_crowded3.err; python - <<PY
import json
d=json.load(open('gpurun_out/r5_crowded3.json'))
print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], {k:(v.get('avg_launch_ms') or v.get('ms_per_step')) for k,v in d['kernels'].items()})
PY
