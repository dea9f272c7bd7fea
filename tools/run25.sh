for L in 48 24 12 6; do KASA_LONG_STEPS=$L python bench.py --crowded --no-pmc --steps 2 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('long steps $L group', d['kernels']['group_kernel']['avg_launch_ms'], 'value', d['value'])"; done
