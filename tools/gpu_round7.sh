#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_gpu.log
for nb in 0 1 2 3 4; do echo "== 1024 f64 nb=$nb"; WV_FUSED_BOUNDARY_BLOCKS=$nb python bench.py --no-cpu-baseline --no-small --steps 100 --warmup 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
echo "== 256"; python bench.py --nx 256 --ny 256 --nz 256 --no-cpu-baseline --no-small --steps 2000 --warmup 200 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
echo "== 512"; for nb in 0 2; do WV_FUSED_BOUNDARY_BLOCKS=$nb python bench.py --nx 512 --ny 512 --nz 512 --no-cpu-baseline --no-small --steps 500 --warmup 50 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fused -o b -- python bench.py --no-cpu-baseline --no-small --steps 30 --warmup 5 > gpurun_out/prof_fused.log 2>&1; head -5 gpurun_out/prof_fused/b_kernel_stats.csv
