#!/bin/bash
# One GPU call (`gpurun -- tools/gpu_call.sh <round> <task> [<task> ...]`): each task writes its evidence under gpurun_out/<round>/ and
# prints a one-line summary; what is to be judged is copied from there into profiles/<round>/.  Tasks:
#   tests[:<pytest selection>]   pytest -m gpu (default: the whole suite) -> pytest_gpu[_N].txt
#   smoke                        __graft_entry__.smoke()
#   bench[:<bench.py args>]      bench.py -> bench[_N].json
#   kernels                      rocprofv3 --kernel-trace --stats of bench.py (short) -> rocprof_kernel_stats.csv
#   pmc                          FETCH_SIZE / WRITE_SIZE passes of bench.py (separate passes, kernel trace only) -> pmc_summary.json
#   slabs[:<worlds>[:<tuning>]]  tools/slab_overhead.py for the given worlds (default 2,4,8), twice each -> slab_overhead_one_gpu.txt
#   timeline[:<tuning>]          rocprofv3 kernel trace (the copy-trace domain crashes rocprofv3 on this image) of 8 x 128 planes -> slab_pass_timeline_8x128[_tuning].txt
#   middle                       tools/middle_rank_bench.py
#   boundary_test                tools/boundary_test_reproduction.py --engine, both sources -> boundary_test_*.{txt,npz}
#   run:<command>                anything else, output to run_N.txt
# (A task is ONE shell word: quote it on the gpurun command line, and name tests by node id -- tests/x.py::test_y -- rather than with
# `-k "a or b"`, whose inner quotes do not survive the trip.)
R=$1; shift
export TMPDIR=/tmp; O=gpurun_out/$R; mkdir -p $O
n=0
for task in "$@"; do
  n=$((n+1)); kind=${task%%:*}; arg=""; [[ "$task" == *:* ]] && arg=${task#*:}
  case $kind in
    tests)
      sel=${arg:-tests}
      ( time timeout 2400 python -m pytest $sel -m gpu -q --durations=10 ) > $O/pytest_gpu_$n.txt 2>&1
      grep -h "passed\|failed\|error" $O/pytest_gpu_$n.txt | tail -3 ;;
    smoke) python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt ;;
    bench)
      python bench.py $arg > $O/bench_$n.json 2> $O/bench_$n.err; tail -1 $O/bench_$n.json | cut -c1-400
      python3 -c "
import json,sys
d=json.loads(open('$O/bench_$n.json').read().strip().splitlines()[-1]); r=d['roofline']
print('  ->', d['value'], d['unit'], '| kernel', r['kernel'], r['kernel_ms'], 'ms frac', r['frac'], '| whole step', r.get('whole_step_frac'), '| 256^3', d['config'].get('also_256cubed_gnode_per_s'), '| halo', d['config'].get('halo_measured'))" 2>/dev/null ;;
    kernels)
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o k -- python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-small --no-reference-on-gpu $arg > $O/ks.log 2>&1
      f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprof_kernel_stats.csv && python tools/kernel_stats.py $O/rocprof_kernel_stats.csv | head -12
      rm -rf $O/ks ;;
    pmc)
      CMD="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-small --no-reference-on-gpu $arg"
      for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o b -- $CMD > $O/pmc_$c.log 2>&1
      done
      python3 tools/pmc_summary.py $O | tee $O/pmc_summary.txt
      rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
    slabs)
      worlds=${arg%%:*}; tun=""; [[ "$arg" == *:* ]] && tun="--tuning ${arg#*:}"; worlds=${worlds:-2,4,8}
      for rep in 1 2; do for w in ${worlds//,/ }; do python tools/slab_overhead.py --world $w $tun 2>&1 | grep fp64; done; done | tee -a $O/slab_overhead_one_gpu.txt ;;
    timeline)
      tun=""; tag=""; [ -n "$arg" ] && tun="--tuning $arg" && tag="_${arg//[=,]/_}"
      rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 $tun > $O/tl.log 2>&1
      python tools/pass_timeline.py $O/tl 4 > $O/slab_pass_timeline_8x128$tag.txt 2>&1; head -3 $O/slab_pass_timeline_8x128$tag.txt; rm -rf $O/tl $O/tl.log ;;
    middle) python tools/middle_rank_bench.py 2>&1 | grep "middle rank" | tee -a $O/middle_rank_bench.txt ;;
    boundary_test)
      for src in transparent soft; do
        ( time python tools/boundary_test_reproduction.py --engine --source $src --save $O/boundary_test_$src.npz ) > $O/boundary_test_$src.txt 2>&1
        grep -A10 "output.$src/" $O/boundary_test_$src.txt | head -11
      done ;;
    run) ( time bash -c "$arg" ) > $O/run_$n.txt 2>&1; tail -5 $O/run_$n.txt ;;
    *) echo "unknown task $task" ;;
  esac
done
