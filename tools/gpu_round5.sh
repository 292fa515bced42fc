#!/bin/bash
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
echo "== bench overlap on"; timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_overlap.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_overlap.log
echo "== bench overlap off"; WV_BOUNDARY_OVERLAP=0 timeout 900 python bench.py --no-cpu-baseline --no-small > gpurun_out/bench_nooverlap.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_nooverlap.log
echo "== cpu baseline"; timeout 900 python bench.py --steps 20 --warmup 5 --no-small > gpurun_out/bench_cpu.log 2>&1; tail -1 gpurun_out/bench_cpu.log | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['cpu_baseline'])"
