#!/bin/bash
# round 3, fourth GPU call: slabs with merged face launches; per-kernel times of a pass; slab overhead; configs[3] figure
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_rccl_chain.py tests/test_gpu_concert.py tests/test_gpu_api.py tests/test_gpu_config3.py tests/test_gpu_bench_world.py -x -q -m gpu ) > $O/pytest_call4.txt 2>&1; tail -6 $O/pytest_call4.txt
for p in f64 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$p -o b -- python bench.py --precision $p --steps 30 --warmup 6 --no-cpu-baseline --no-small --no-reference-on-gpu > $O/bench_$p.json 2> $O/bench_$p.err
  echo "$p: $(cut -c1-110 $O/bench_$p.json)"; python tools/kernel_stats.py $O/tr_$p pair_march boundary_kernel xwall; cp $O/tr_$p/*/*kernel_stats.csv $O/kernel_stats_$p.csv 2>/dev/null || cp $(find $O/tr_$p -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$p.csv; rm -rf $O/tr_$p
done 2>&1 | tee $O/pass_kernels.txt
for w in 2 8; do python tools/slab_overhead.py --world $w --steps 40; done 2>&1 | tee $O/slab_overhead_one_gpu.txt
python tools/config3_one_gpu.py 2>&1 | tee $O/config3_one_gpu.txt
