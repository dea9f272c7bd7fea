python -m pytest tests/test_gpu_partition.py -x -q -m gpu -k "bench" > gpurun_out/r5_t12.log 2>&1; tail -3 gpurun_out/r5_t12.log
python3 tools/make_profiles.py r05 > gpurun_out/r5_make_profiles.log 2>&1; tail -5 gpurun_out/r5_make_profiles.log
