#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
echo "== stream_bench"; timeout 900 tools/stream_bench 1024 20 > gpurun_out/stream_bench_1024.jsonl 2>&1; echo "rc=$?"; grep -v '"base"' gpurun_out/stream_bench_1024.jsonl | grep -v '"copy' 
echo "== pmc"; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1 -o p -- tools/stream_bench 1024 5 prof > gpurun_out/pmc1.log 2>&1; echo "rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc2 -o p -- tools/stream_bench 1024 5 prof > gpurun_out/pmc2.log 2>&1
python3 - <<'PY'
import csv, collections
for d in ('pmc1','pmc2'):
    rows=list(csv.DictReader(open('gpurun_out/%s/p_counter_collection.csv'%d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][5:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'march' in k or 'triad' in k:
            print(d,k,{c:"%.4g"%(sum(x)/len(x)) for c,x in v.items()})
PY
