python tools/ab_probe.py --flags 0 --rank-flags "" --rounds 2 --steps 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('flags',d['flags'],'group',d['kernel_ms']['group_kernel'],'tiles',d['group_tiles'])"
tools/asan_run.sh python tools/fuzz_gpu.py 905000 100000 420 > gpurun_out/r5_fuzz_asan2.log 2>&1; tail -6 gpurun_out/r5_fuzz_asan2.log
tools/asan_run.sh python -m pytest tests/test_gpu_cpp_host.py tests/test_batches.py -x -q -m gpu > gpurun_out/r5_asan_cpp.log 2>&1; tail -5 gpurun_out/r5_asan_cpp.log
