python -m pytest tests/test_gpu_parity.py tests/test_gpu_partition.py -x -q -m gpu -k "wide or 128 or full_size or random_conf or k_ranges or synthetic_part" > gpurun_out/r5_t14.log 2>&1; tail -3 gpurun_out/r5_t14.log
for F in 0 16777216 150994944; do python bench.py --wide --no-pmc --steps 3 --warmup 1 --no-cpu --no-e2e --debug-flags $F 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('wide flags $F', d['value'], d['ms_per_step'], 'group', d['kernels']['group_kernel']['avg_launch_ms'], d['batch']['group_tiles_listed'])"; done
