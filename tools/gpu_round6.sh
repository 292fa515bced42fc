#!/bin/bash
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== bench f64"; python bench.py --no-cpu-baseline 2>/dev/null | tail -1
echo "== bench f32"; python bench.py --precision f32 --no-cpu-baseline --no-small --steps 100 --warmup 10 2>/dev/null | tail -1
echo "== 256 f64"; python bench.py --nx 256 --ny 256 --nz 256 --no-cpu-baseline --no-small --steps 2000 --warmup 200 2>/dev/null | tail -1
echo "== 1000^3 f64 (ragged)"; python bench.py --nx 1000 --ny 1000 --nz 1000 --no-cpu-baseline --no-small --steps 50 --warmup 10 2>/dev/null | tail -1
