#!/bin/bash
# round 3, eighth GPU call: where a real room's step goes -- the concert hall meshed for 1600 Hz and 2400 Hz, kernels of its passes
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hall -o h -- python tools/concert_bench.py 1600 > $O/hall.log 2>&1; grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/hall.log | tail -6
python tools/kernel_stats.py $O/hall pair_march boundary_kernel sweep xwall fixup pre_post | tee $O/concert_hall_kernels.txt
rm -rf $O/hall
