python -m pytest tests -x -q -m gpu > gpurun_out/r5_t8.log 2>&1; tail -3 gpurun_out/r5_t8.log
python bench.py --crowded --no-pmc --steps 2 --warmup 2 > gpurun_out/r5_crowded4.json 2> gpurun_out/r5_crowded4.err; python - <<PY
import json
d=json.load(open('gpurun_out/r5_crowded4.json'))
print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], {k:(v.get('avg_launch_ms') or v.get('ms_per_step')) for k,v in d['kernels'].items()})
PY
