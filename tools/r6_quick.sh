#!/bin/bash
# round 6: quick parity subset (+ optional A/B of debug flags on the bench batch: FLAGS=a,b)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-adversarial or random_conf or general_kernel or medium or golden_files or many_taxa or long_reads or unique_drops or records_after or very_long}" > gpurun_out/r6_quick_tests.log 2>&1
tail -15 gpurun_out/r6_quick_tests.log
if [ -n "$FLAGS" ]; then
timeout 900 python tools/ab_probe.py --flags $FLAGS --rank-flags "" --rounds 2 --steps 3 > gpurun_out/r6_ab.log 2>&1
tail -6 gpurun_out/r6_ab.log
fi
