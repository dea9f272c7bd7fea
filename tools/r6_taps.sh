#!/bin/bash
# round 6: where group2_kernel's time goes -- phase taps (KASA_G2_TAP 32: A; 64: A + B; 256: + allocation; 128: everything but the
# keys) and timing variants (KASA_G2_VAR), on the bench batch: 64-byte cells (flags 0) and 32-byte slots (134217728)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r6_taps.log
for cfg in ${CFGS:-"0:0 256:0"}; do
  tap=${cfg%%:*}; var=${cfg##*:}
  echo "== KASA_G2_TAP=$tap KASA_G2_VAR=$var" >> gpurun_out/r6_taps.log
  KASA_G2_TAP=$tap KASA_G2_VAR=$var timeout 600 python tools/ab_probe.py --flags ${FLAGS:-0,134217728} --rank-flags "" --rounds 1 --steps 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: o = json.loads(l)
    except Exception: continue
    print(o['flags'], 'group_kernel', o['kernel_ms'].get('group_kernel'), 'wall', o['ms_per_step_wall'])
" >> gpurun_out/r6_taps.log
done
cat gpurun_out/r6_taps.log
