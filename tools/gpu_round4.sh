#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== sweep"; timeout 900 python tools/sweep_stream.py --out gpurun_out/sweep_1024_f64.json --steps 10 > gpurun_out/sweep.log 2>&1; echo "rc=$?"; grep -E "BEST|mesh" gpurun_out/sweep.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log
echo "== pmc"; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1 -o p -- tools/stream_bench 1024 5 prof > gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc2 -o p -- tools/stream_bench 1024 5 prof > gpurun_out/pmc2.log 2>&1
python3 - <<'PY'
import csv, collections
for d in ('pmc1','pmc2'):
    rows=list(csv.DictReader(open('gpurun_out/%s/p_counter_collection.csv'%d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][5:64]+" grid="+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'march' in k or 'triad' in k or 'sweep' in k:
            print(d,k,{c:"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
