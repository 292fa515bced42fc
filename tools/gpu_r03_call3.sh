#!/bin/bash
# round 3, third GPU call: x-facing walls on compact copies -- parity suites, then kernel times A/B at 1024^3
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_pair.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api_sequences.py tests/test_gpu_slabs.py tests/test_gpu_rccl_chain.py tests/test_gpu_concert.py tests/test_tile_lists.py tests/test_gpu_api.py -x -q -m gpu ) > $O/pytest_call3.txt 2>&1; tail -25 $O/pytest_call3.txt
for x in 1 0; do for p in f64 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$x$p -o b -- python bench.py --precision $p --tuning boundary_xwall=$x --steps 30 --warmup 6 --no-cpu-baseline --no-small --no-reference-on-gpu > $O/xwall_${x}_$p.json 2> $O/xwall_${x}_$p.err
  echo "boundary_xwall=$x $p: $(cut -c1-120 $O/xwall_${x}_$p.json)"
  grep -h "boundary_kernel\|pair_march\|xwall" $O/tr_$x$p/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
  rm -rf $O/tr_$x$p
done; done 2>&1 | tee $O/xwall_ab.txt
