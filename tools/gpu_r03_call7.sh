#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 > $O/tl.log 2>&1; tail -1 $O/tl.log
python tools/pass_timeline.py $O/tl 4 > $O/slab_pass_timeline.txt 2>&1; head -120 $O/slab_pass_timeline.txt
rm -rf $O/tl
