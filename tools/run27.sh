python -m pytest tests -x -q -m gpu > gpurun_out/r5_t16.log 2>&1; tail -3 gpurun_out/r5_t16.log
python tools/fuzz_gpu.py 940000 100000 120 > gpurun_out/r5_fuzz6.log 2>&1; tail -1 gpurun_out/r5_fuzz6.log
python3 tools/make_profiles.py r05 > gpurun_out/r5_make_profiles3.log 2>&1; tail -1 gpurun_out/r5_make_profiles3.log | cut -c1-200
python3 tools/make_profiles.py r05 --wide > gpurun_out/r5_make_profiles_wide3.log 2>&1; tail -2 gpurun_out/r5_make_profiles_wide3.log
python3 tools/make_profiles.py r05 --crowded > gpurun_out/r5_make_profiles_crowded3.log 2>&1; tail -8 gpurun_out/r5_make_profiles_crowded3.log
