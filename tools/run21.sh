python -m pytest tests -x -q -m gpu > gpurun_out/r5_t13.log 2>&1; tail -3 gpurun_out/r5_t13.log
python tools/fuzz_gpu.py 930000 100000 150 > gpurun_out/r5_fuzz5.log 2>&1; tail -1 gpurun_out/r5_fuzz5.log
python bench.py --wide --no-pmc --steps 3 --warmup 1 --no-cpu --no-e2e 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('wide', d['value'], d['ms_per_step'], d['stage_ms_per_step'], {k:round(v.get('avg_launch_ms',0) or v.get('ms_per_step',0),2) for k,v in d['kernels'].items()}, d['batch'])"
python bench.py --crowded --no-pmc --steps 2 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], {k:round(v.get('avg_launch_ms',0) or v.get('ms_per_step',0),2) for k,v in d['kernels'].items()})"
python tools/ab_probe.py --flags 0 --rank-flags "" --rounds 1 --steps 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 group', d['kernel_ms']['group_kernel'], d['stage_ms'])"
