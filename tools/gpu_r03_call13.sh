#!/bin/bash
# (a) the GPU suite (minus its heaviest cases) with the library's host code under ASan + UBSan; (b) why do FOUR slabs cost more
# than eight?  timeline of a pass of 4 x 256 planes
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
(timeout 600 tools/sanitizer_run.sh tests -q -x -m gpu -k "not 1024 and not config3 and not div3 and not long and not 10000 and not full_size and not bench_world" 2>&1 | tail -40) > $O/sanitizers_gpu_suite.txt; tail -6 $O/sanitizers_gpu_suite.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 4 --steps 8 > $O/tl.log 2>&1; tail -1 $O/tl.log
python tools/pass_timeline.py $O/tl 4 > $O/slab_pass_timeline_4x256.txt 2>&1; rm -rf $O/tl $O/tl.log
