#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
python -m pytest tests -m gpu -x -q > gpurun_out/ab/pytest.txt 2>&1; tail -2 gpurun_out/ab/pytest.txt
for o in 0 1; do
  WV_BOUNDARY_ORDER=$o rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/t$o -o b -- \
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-small > gpurun_out/ab/log$o.txt 2>&1
  echo "order=$o"; grep -h "boundary_kernel\|sweep" gpurun_out/ab/t$o/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
  rm -rf gpurun_out/ab/t$o
done
