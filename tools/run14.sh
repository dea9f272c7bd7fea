python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "huge_taxon or full_size or random_conf" > gpurun_out/r5_t9.log 2>&1; tail -2 gpurun_out/r5_t9.log
python bench.py --crowded --no-pmc --steps 2 --warmup 2 > gpurun_out/r5_crowded5.json 2> gpurun_out/r5_crowded5.err; python - <<PY
import json
d=json.load(open('gpurun_out/r5_crowded5.json'))
print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], {k:(v.get('avg_launch_ms') or v.get('ms_per_step')) for k,v in d['kernels'].items()})
PY
