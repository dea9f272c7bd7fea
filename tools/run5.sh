python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5_t4.log 2>&1; tail -3 gpurun_out/r5_t4.log
for G in 0 3 2 6; do KASA_G2_GRID=$G python tools/ab_probe.py --flags 0 --rank-flags "" --rounds 1 --steps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $G', d['kernel_ms']['group_kernel'], d['stage_ms'])"; done
python bench.py --crowded --no-pmc --steps 2 --warmup 2 > gpurun_out/r5_crowded1.json 2> gpurun_out/r5_crowded1.err; python - <<PY
import json
d=json.load(open('gpurun_out/r5_crowded1.json'))
print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['kernels'], d['batch'])
PY
python tools/fuzz_gpu.py 910000 100000 240 > gpurun_out/r5_fuzz1.log 2>&1; tail -2 gpurun_out/r5_fuzz1.log
tools/asan_run.sh python tools/fuzz_gpu.py 905000 100000 300 > gpurun_out/r5_fuzz_asan1.log 2>&1; tail -5 gpurun_out/r5_fuzz_asan1.log
