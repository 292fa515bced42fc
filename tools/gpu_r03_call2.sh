#!/bin/bash
# round 3, second GPU call: configs[3], the chain tests (agreements, shm stand-in), bench.py with world > 1
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_config3.py -x -q -m gpu ) > $O/pytest_config3.txt 2>&1; tail -4 $O/pytest_config3.txt
( time timeout 900 python -m pytest tests/test_gpu_rccl_chain.py tests/test_gpu_bench_world.py tests/test_gpu_slabs.py tests/test_gpu_api.py tests/test_gpu_concert.py tests/test_cpp_api.py -x -q -m gpu ) > $O/pytest_chain_call2.txt 2>&1; tail -30 $O/pytest_chain_call2.txt
