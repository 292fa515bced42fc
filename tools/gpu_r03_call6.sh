#!/bin/bash
# round 3, sixth GPU call: timeline of a slab pass; the march's instruction-only time; single steps at 640^3 / 768^3; 256^3
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
python -c "from wayverb_amd import build; build.build(verbose=False)"
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 > $O/tl.log 2>&1; tail -1 $O/tl.log
python tools/pass_timeline.py $O/tl 5 2>&1 | tee $O/slab_pass_timeline.txt | head -80
rm -rf $O/tl
tools/pair_tune 1024 4 2>&1 | tee $O/pair_tune_r03.txt | head -14
B="python bench.py --no-cpu-baseline --no-small --no-reference-on-gpu"
for n in 512 640 768; do echo "n=$n single steps: $($B --tuning pair=0 --nx $n --ny $n --nz $n --steps 300 --warmup 30 | cut -c1-100)"; done 2>&1 | tee $O/single_steps_by_size.txt
for p in -1 0; do echo "256^3 pair=$p: $($B --tuning pair=$p --nx 256 --ny 256 --nz 256 --steps 4000 --warmup 200 | cut -c1-100)"; done 2>&1 | tee $O/bench_256.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr256 -o b -- $B --nx 256 --ny 256 --nz 256 --steps 600 --warmup 20 > /dev/null 2>&1; python tools/kernel_stats.py $O/tr256 pair_march boundary_kernel | tee -a $O/bench_256.txt; rm -rf $O/tr256
