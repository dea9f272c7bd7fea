python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "huge_taxon or full_size" > gpurun_out/r5_t6.log 2>&1; tail -3 gpurun_out/r5_t6.log
python tools/fuzz_gpu.py 911700 100000 120 > gpurun_out/r5_fuzz3.log 2>&1; tail -2 gpurun_out/r5_fuzz3.log
python bench.py --crowded --no-pmc --steps 2 --warmup 2 > gpurun_out/r5_crowded2.json 2> gpurun_out/r5_crowded2.err; python - <<PY
import json
d=json.load(open('gpurun_out/r5_crowded2.json'))
print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], {k:(v.get('avg_launch_ms') or v.get('ms_per_step')) for k,v in d['kernels'].items()}, d['batch'])
PY
