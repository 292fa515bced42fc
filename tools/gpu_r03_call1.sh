#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite on the split engine + configs[3] at full size + a bench line
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_config3.py -x -q -m gpu ) > $O/pytest_config3.txt 2>&1; tail -5 $O/pytest_config3.txt
( time timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_config3.py ) > $O/pytest_gpu_call1.txt 2>&1; tail -5 $O/pytest_gpu_call1.txt
timeout 600 python bench.py --steps 40 --warmup 10 --no-reference-on-gpu --cpu-seconds 4 > $O/bench_call1.json 2> $O/bench_call1.err; tail -c 1500 $O/bench_call1.json
rocm-smi --showmeminfo vram | head -8
