python3 tools/make_profiles.py r05 --wide > gpurun_out/r5_make_profiles_wide.log 2>&1; tail -8 gpurun_out/r5_make_profiles_wide.log
python3 tools/make_profiles.py r05 --crowded > gpurun_out/r5_make_profiles_crowded.log 2>&1; tail -8 gpurun_out/r5_make_profiles_crowded.log
