#!/bin/bash
# round-3 closing run on the final host code (kernels unchanged since profiles/traffic.json was measured): the whole GPU suite, smoke,
# bench.py, 1024^3 as slabs on one GPU (the engine's choice, and a thin slab's march forced into one round), the middle rank of configs[3]
R=r03; export TMPDIR=/tmp; O=gpurun_out/$R; mkdir -p $O
( time python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest_gpu.txt 2>&1; grep -h "passed\|failed" $O/pytest_gpu.txt | tail -1
python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -1 $O/bench_n1.json | cut -c1-300
(for rep in 1 2; do for w in 2 4 8; do python tools/slab_overhead.py --world $w 2>&1 | grep fp64; done; done
 for rep in 1 2; do for w in 4 8; do python tools/slab_overhead.py --world $w --tuning pair_chunks=1 2>&1 | grep fp64; done; done) | tee $O/slab_overhead_one_gpu.txt
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 > $O/tl.log 2>&1
python tools/pass_timeline.py $O/tl 4 > $O/slab_pass_timeline_8x128.txt 2>&1; rm -rf $O/tl $O/tl.log
python tools/middle_rank_bench.py 2>&1 | grep "middle rank" | tee $O/middle_rank_bench.txt
