python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "huge_taxon or full_size or adversarial" > gpurun_out/r5_t11.log 2>&1; tail -2 gpurun_out/r5_t11.log
python bench.py --crowded --no-pmc --steps 2 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('crowded', d['value'], d['ms_per_step'], d['stage_ms_per_step'], 'group', d['kernels']['group_kernel']['avg_launch_ms'], 'dense', d['kernels']['score_dense_kernel'])"
