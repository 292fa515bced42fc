#!/bin/bash
# rocprofv3 kernel-trace stats + PMC traffic for the bench workload; summaries go to gpurun_out/prof
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
CMD="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-small"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o bench -- $CMD > gpurun_out/prof/trace.log 2>&1; echo "trace rc=$?"
tail -1 gpurun_out/prof/trace.log
find gpurun_out/prof/trace -name "*stats*.csv" | head; 
for f in $(find gpurun_out/prof/trace -name "*kernel_stats.csv"); do cat $f | head -12; done
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -o bench -- $CMD > gpurun_out/prof/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -o bench -- $CMD > gpurun_out/prof/pmc_write.log 2>&1; echo "pmc write rc=$?"
python3 - <<'PY'
import csv, collections, glob, json
out={}
for tag in ('pmc_fetch','pmc_write'):
    for f in glob.glob('gpurun_out/prof/%s/**/*counter_collection.csv'%tag, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            for c,x in v.items():
                out.setdefault(k,{})[c]={'mean':sum(x)/len(x),'n':len(x)}
json.dump(out, open('gpurun_out/prof/pmc_summary.json','w'), indent=1)
for k,v in out.items():
    if 'stream' in k or 'boundary' in k: print(k, v)
PY
