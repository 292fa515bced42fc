#!/bin/bash
# Boundary-kernel time as a function of which faces dominate: three boxes with the same node count.
export TMPDIR=/tmp; mkdir -p gpurun_out/shapes
for dims in "2048 1024 512" "512 1024 2048" "1024 1024 1024" "1024 2048 512" "1024 512 2048"; do
  set -- $dims
  tag="$1x$2x$3"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shapes/$tag -o b -- \
    python bench.py --nx $1 --ny $2 --nz $3 --steps 30 --warmup 5 --no-cpu-baseline --no-small > gpurun_out/shapes/$tag.log 2>&1
  echo "== $tag"; grep -h "sweep\|boundary_kernel" gpurun_out/shapes/$tag/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
  rm -rf gpurun_out/shapes/$tag
done
