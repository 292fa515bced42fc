#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
python tools/concert_bench.py 800 1600 2400 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $O/concert_hall_steps.txt
timeout 900 python -m pytest tests/test_tile_lists.py tests/test_gpu_concert.py tests/test_gpu_pair.py tests/test_gpu_fuzz.py tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | grep "passed\|failed"
python tools/room_bench.py 768 2>&1 | grep "Gnode" | tee $O/room_bench.txt
