#!/bin/bash
# Round-end evidence: tests, rocprofv3 kernel stats + PMC traffic, bench (N=1), engine sweeps.
R=${1:-r02}
mkdir -p gpurun_out/$R; export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/$R/pytest_gpu.txt 2>&1; grep -h "passed\|failed" gpurun_out/$R/pytest_gpu.txt | tail -1
python __graft_entry__.py --smoke > gpurun_out/$R/smoke.txt 2>&1; tail -1 gpurun_out/$R/smoke.txt
CMD="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-small --no-reference-on-gpu"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace -o bench -- $CMD > gpurun_out/$R/trace.log 2>&1
cp gpurun_out/$R/trace/bench_kernel_stats.csv gpurun_out/$R/rocprof_kernel_stats.csv; head -4 gpurun_out/$R/rocprof_kernel_stats.csv | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmc_fetch -o bench -- $CMD > gpurun_out/$R/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmc_write -o bench -- $CMD > gpurun_out/$R/pmc_write.log 2>&1
python3 - "$R" <<'PY'
import csv, collections, glob, json, sys
sys.path.insert(0, ".")
import bench
R=sys.argv[1]; out={}
for tag in ('pmc_fetch','pmc_write'):
    for f in glob.glob('gpurun_out/%s/%s/**/*counter_collection.csv'%(R,tag), recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            for c,x in v.items(): out.setdefault(k,{})[c]={'mean':sum(x)/len(x),'n':len(x)}
json.dump(out, open('gpurun_out/%s/pmc_summary.json'%R,'w'), indent=1)
for k,v in out.items():
    if 'pair' in k or 'sweep' in k or 'boundary_kernel' in k: print(k, v)
    if 'pair_march_kernel<double' in k and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        rec = {"workload": "1024x1024x1024 f64", "kernel": "pair_march_kernel", "kernel_full_name": k.replace('void ', ''),
               "kernel_sources": bench.kernel_sources_hash(), "measured": R, "files": "%s/pmc_summary.json" % R,
               "source": "profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, launches of `bench.py --steps 30 --warmup 6`)" % R,
               "FETCH_SIZE_KiB_raw": v['FETCH_SIZE']['mean'], "WRITE_SIZE_KiB_raw": v['WRITE_SIZE']['mean'],
               "corrections": "gfx950: FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> x2 (calibrated on the in-place triad in tools/stream_bench.hip: 2*FETCH_SIZE*1024 = bytes read, exactly); WRITE_SIZE*1024 = bytes written, exactly; Infinity-Cache hits are counted (fabric-side counter)",
               "hbm_bytes_per_launch": int(2 * v['FETCH_SIZE']['mean'] * 1024 + v['WRITE_SIZE']['mean'] * 1024)}
        json.dump(rec, open('gpurun_out/%s/traffic.json' % R, 'w'), indent=1)
        json.dump(rec, open('profiles/traffic.json', 'w'), indent=1)   # what this box's bench run reports
PY
rm -rf gpurun_out/$R/trace gpurun_out/$R/pmc_fetch gpurun_out/$R/pmc_write
python bench.py > gpurun_out/$R/bench_n1.json 2> gpurun_out/$R/bench_n1.err; tail -1 gpurun_out/$R/bench_n1.json | cut -c1-400
python bench.py --precision f32 --no-cpu-baseline --no-small > gpurun_out/$R/bench_n1_f32.json 2>/dev/null
python bench.py --nx 256 --ny 256 --nz 256 --no-cpu-baseline --no-small --steps 10000 --warmup 500 > gpurun_out/$R/bench_256cubed_10k_steps.json 2>/dev/null; tail -1 gpurun_out/$R/bench_256cubed_10k_steps.json | cut -c1-200
python bench.py --nx 1000 --ny 1000 --nz 1000 --no-cpu-baseline --no-small > gpurun_out/$R/bench_1000cubed.json 2>/dev/null

python bench.py --tuning pair=0 --no-cpu-baseline --no-small --no-reference-on-gpu > gpurun_out/$R/bench_n1_single_steps_only.json 2>/dev/null; tail -1 gpurun_out/$R/bench_n1_single_steps_only.json | cut -c1-200
for n in 128 160 256 384 512 768; do for p in 0 1; do echo "n=$n WV_PAIR=$p: $(python bench.py --tuning pair=$p --nx $n --ny $n --nz $n --no-cpu-baseline --no-small --no-reference-on-gpu --steps 400 --warmup 40 2>/dev/null | cut -c1-140)"; done; done > gpurun_out/$R/pair_vs_single_by_size.txt
tools/pair_tune 1024 6 > gpurun_out/$R/pair_tune.txt 2>&1
ls gpurun_out/$R
python tools/middle_rank_bench.py > gpurun_out/$R/middle_rank_bench.txt 2>&1; tail -2 gpurun_out/$R/middle_rank_bench.txt
python tools/concert_bench.py > gpurun_out/$R/concert_hall_steps.txt 2>&1; tail -8 gpurun_out/$R/concert_hall_steps.txt
