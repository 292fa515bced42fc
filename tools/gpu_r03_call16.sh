#!/bin/bash
# a slab's march in at least two rounds of workgroups (so that the exchange gets a CU before it ends): chains still exact?  cost on one GPU?
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_rccl_chain.py tests/test_gpu_concert.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tee $O/slab_tests_call16.txt
for rep in 1 2; do for w in 2 4 8; do python tools/slab_overhead.py --world $w 2>&1 | grep fp64; done; done | tee $O/slab_overhead_two_rounds.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 > $O/tl.log 2>&1; tail -1 $O/tl.log
python tools/pass_timeline.py $O/tl 4 > $O/slab_pass_timeline_8x128_two_rounds.txt 2>&1; rm -rf $O/tl $O/tl.log
