for L in 3 4 8; do KASA_LONG_STEPS=$L python bench.py --crowded --no-pmc --steps 2 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('crowded long steps $L group', d['kernels']['group_kernel']['avg_launch_ms'], 'value', d['value'])"; done
for L in 48 12 6; do KASA_LONG_STEPS=$L python tools/ab_probe.py --flags 0 --rank-flags "" --rounds 1 --steps 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 long steps $L group', d['kernel_ms']['group_kernel'])"; done
