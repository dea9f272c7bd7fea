python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "huge_taxon or full_size" > gpurun_out/r5_t10.log 2>&1; tail -2 gpurun_out/r5_t10.log
python tools/fuzz_gpu.py 920000 100000 90 > gpurun_out/r5_fuzz4.log 2>&1; tail -1 gpurun_out/r5_fuzz4.log
python bench.py --crowded --no-pmc --steps 2 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], 'dense', d['kernels']['score_dense_kernel'])"
