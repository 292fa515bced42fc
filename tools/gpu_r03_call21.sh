#!/bin/bash
# the wall filter design changed (IT++'s index arithmetic): the GPU tests that use designed filters, and mic_test with the engine stepping
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
timeout 100 python -m pytest tests/test_mic_test_reference.py tests/test_gpu_concert.py tests/test_cpp_api.py -q -m gpu -x 2>&1 | grep "passed\|failed\|Error\|worst" | tee $O/designed_filter_tests.txt
