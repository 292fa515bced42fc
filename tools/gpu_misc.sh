#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== f32 1024^3"; python bench.py --precision f32 --no-cpu-baseline --no-small --steps 100 --warmup 10 | tail -1
echo "== f64 256^3 direct"; python bench.py --nx 256 --ny 256 --nz 256 --no-cpu-baseline --no-small --steps 2000 --warmup 200 | tail -1
echo "== rocprof 256^3"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof256 -o b -- python bench.py --nx 256 --ny 256 --nz 256 --no-cpu-baseline --no-small --steps 500 --warmup 50 > gpurun_out/prof256.log 2>&1; head -8 gpurun_out/prof256/b_kernel_stats.csv
echo "== cpu baseline"; python bench.py --steps 10 --warmup 2 --no-small | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['cpu_baseline'])"
