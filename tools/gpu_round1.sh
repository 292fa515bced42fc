#!/bin/bash
# First GPU pass: smoke, parity tests, stream-kernel sweep, bench, rocprof kernel trace.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; free -g | head -2; nproc) > gpurun_out/box.txt 2>&1
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== sweep"; timeout 900 python tools/sweep_stream.py --out gpurun_out/sweep_1024_f64.json > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"; tail -4 gpurun_out/sweep.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench.log
echo "== rocprof"; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o bench -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-small > gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"; tail -2 gpurun_out/rocprof.log
ls -R gpurun_out | head -40
