#!/bin/bash
# slabs of one device take turns at the march: chains still exact?  1024^3 as 2 / 4 / 8 slabs, twice each; timeline of 8 x 128 planes;
# then the GPU suite (minus its heaviest cases and the device-memory leak check: ASan's quarantine keeps freed device memory)
# with the library's host code under ASan + UBSan
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_rccl_chain.py tests/test_gpu_config3.py tests/test_gpu_concert.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tee $O/slab_tests_call15.txt
for rep in 1 2; do for w in 2 4 8; do python tools/slab_overhead.py --world $w 2>&1 | grep fp64; done; done | tee $O/slab_overhead_one_gpu.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 > $O/tl.log 2>&1; tail -1 $O/tl.log
python tools/pass_timeline.py $O/tl 4 > $O/slab_pass_timeline_8x128.txt 2>&1; rm -rf $O/tl $O/tl.log
(timeout 700 tools/sanitizer_run.sh tests -q -x -m gpu -k "not 1024 and not config3 and not div3 and not long and not 10000 and not full_size and not bench_world and not leak" 2>&1 | tail -60) > $O/sanitizers_gpu_suite.txt; tail -8 $O/sanitizers_gpu_suite.txt
