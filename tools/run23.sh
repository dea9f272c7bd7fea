python -m pytest tests -x -q -m gpu > gpurun_out/r5_t15.log 2>&1; tail -3 gpurun_out/r5_t15.log
python3 tools/make_profiles.py r05 > gpurun_out/r5_make_profiles2.log 2>&1; tail -2 gpurun_out/r5_make_profiles2.log | cut -c1-300
python3 tools/make_profiles.py r05 --wide > gpurun_out/r5_make_profiles_wide2.log 2>&1; tail -3 gpurun_out/r5_make_profiles_wide2.log
python3 tools/make_profiles.py r05 --crowded > gpurun_out/r5_make_profiles_crowded2.log 2>&1; tail -3 gpurun_out/r5_make_profiles_crowded2.log
tools/asan_run.sh python tools/fuzz_gpu.py 1080000 100000 560 > gpurun_out/r5_fuzz_asan_tail.log 2>&1; tail -2 gpurun_out/r5_fuzz_asan_tail.log
