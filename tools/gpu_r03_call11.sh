#!/bin/bash
# the local transport waits per field buffer / per step parity instead of for its neighbours' latest events:
# chains still exact?  what do 2 / 8 slabs of 1024^3 cost now, with 4 hardware queues and with 16?
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_rccl_chain.py tests/test_gpu_config3.py -x -q -m gpu 2>&1 | grep "passed\|failed\|Error" | tee $O/slab_tests_call11.txt
for w in 2 8; do
  python tools/slab_overhead.py --world $w 2>&1 | grep fp64
  python tools/slab_overhead.py --world $w --hw-queues 16 2>&1 | grep fp64 | sed 's/^/hw-queues 16: /'
done | tee $O/slab_overhead_call11.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 --hw-queues 16 > $O/tl.log 2>&1; tail -1 $O/tl.log
python tools/pass_timeline.py $O/tl 4 > $O/slab_pass_timeline_8x128_call11.txt 2>&1; rm -rf $O/tl
