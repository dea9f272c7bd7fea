for T in 0 1 2 4 7; do KASA_DENSE_TAP=$T python bench.py --crowded --no-pmc --steps 2 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('tap $T dense', d['kernels']['score_dense_kernel'], 'score', d['stage_ms_per_step']['score'])"; done
