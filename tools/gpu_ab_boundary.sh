#!/bin/bash
# A/B of one boundary-kernel switch (wv_tuning field name in $1, values 0 and 1) with rocprofv3 kernel stats.
export TMPDIR=/tmp; mkdir -p gpurun_out/ab; VAR=${1:-boundary_lds}
for o in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/t$o -o b -- \
    python bench.py --tuning $VAR=$o --steps 30 --warmup 5 --no-cpu-baseline --no-small > gpurun_out/ab/log$o.txt 2>&1
  echo "$VAR=$o"; grep -h "boundary_kernel\|sweep" gpurun_out/ab/t$o/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
  rm -rf gpurun_out/ab/t$o
done
