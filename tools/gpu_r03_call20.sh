#!/bin/bash
# wait_ghosts waits for the latest of the slab's own pushes (one event instead of eight): every chain test again, the race test, a minute of random chains
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_rccl_chain.py tests/test_gpu_config3.py tests/test_gpu_concert.py tests/test_cpp_api.py tests/test_gpu_bench_world.py -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tee $O/slab_tests_final.txt
timeout 200 python tools/extended_fuzz.py --first 20000 --count 3000 --seconds 60 --only "slab chains" 2>&1 | grep "FAILED\|passed\|Error" | tee -a $O/slab_tests_final.txt
