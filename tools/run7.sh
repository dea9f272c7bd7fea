python tools/fuzz_gpu.py 911776 1 100 2>&1 | tail -1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "huge_taxon or full_size" > gpurun_out/r5_t5.log 2>&1; tail -3 gpurun_out/r5_t5.log
python tools/fuzz_gpu.py 911000 100000 200 > gpurun_out/r5_fuzz2.log 2>&1; tail -2 gpurun_out/r5_fuzz2.log
bash tools/crowded_pmc.sh > gpurun_out/crowded_pmc.log 2>&1; cat gpurun_out/crowded_pmc.log
