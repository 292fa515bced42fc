#!/bin/bash
# the whole GPU suite, smoke and bench.py on the round's final tree
R=r03; export TMPDIR=/tmp; O=gpurun_out/$R; mkdir -p $O
( time python -m pytest tests -m gpu -q --durations=15 ) > $O/pytest_gpu.txt 2>&1; grep -h "passed\|failed" $O/pytest_gpu.txt | tail -1
python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_n1_final_tree.json 2> $O/bench_n1.err; tail -1 $O/bench_n1_final_tree.json | cut -c1-200
