#!/bin/bash
# A/B of a kasa_ctx_debug bit on the bench workload, alternating:  bash tools/ab_flags.sh [bit] [kernel]
BIT=${1:-1024}; K=${2:-group_kernel}
for f in 0 $BIT 0 $BIT 0 $BIT; do python bench.py --steps 4 --warmup 1 --no-cpu --no-e2e --no-secondary --debug-flags $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags', sys.argv[1], round(d['ms_per_step'],1), sys.argv[2], round(d['kernels'][sys.argv[2]]['avg_launch_ms'],2), {k:round(v,1) for k,v in d['stage_ms_per_step'].items()})" $f $K; done
