#!/bin/bash
# round 3, ninth GPU call: the march's unit lists chunk by chunk (neighbouring strips together) against strip by strip, on the concert hall
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
for b in 0 1; do echo "pair_units_by_chunk=$b"; WV_PAIR_UNITS_BY_CHUNK=$b python tools/concert_bench.py 800 1600 2>&1 | grep "two-step passes\|cutoff"; done | tee $O/units_by_chunk.txt
( time timeout 900 python -m pytest tests/test_tile_lists.py tests/test_gpu_concert.py tests/test_gpu_pair.py tests/test_gpu_fuzz.py -x -q -m gpu ) 2>&1 | tail -4
python tools/room_bench.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee $O/room_bench.txt
