for F in 0 32768 65536; do python bench.py --crowded --no-pmc --steps 2 --warmup 2 --debug-flags $F 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('crowded flags $F group', d['kernels']['group_kernel']['avg_launch_ms'], 'stage', d['stage_ms_per_step']['group'])"; done
