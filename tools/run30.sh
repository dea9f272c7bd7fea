python -m pytest tests/test_gpu_cpp_host.py -x -q -k "partitioned" --durations=5 2>&1 | tail -30
