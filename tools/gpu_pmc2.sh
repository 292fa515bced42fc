#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
tools/stream_bench 1024 10 prof2 > gpurun_out/prof2_times.jsonl; cat gpurun_out/prof2_times.jsonl
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcA -o p -- tools/stream_bench 1024 3 prof2 > gpurun_out/pmcA.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/pmcB -o p -- tools/stream_bench 1024 3 prof2 > gpurun_out/pmcB.log 2>&1
python3 - <<'PY'
import csv, collections
for d in ('pmcA','pmcB'):
    rows=list(csv.DictReader(open('gpurun_out/%s/p_counter_collection.csv'%d)))
    # keep dispatch order, group by consecutive kernel runs of 6 launches (3 warm + 3 timed)
    seq=[]
    for r in rows:
        if 'sweep' not in r['Kernel_Name']: continue
        seq.append((int(r['Dispatch_Id']), r['Kernel_Name'][10:52], r['Counter_Name'], float(r['Counter_Value'])))
    by=collections.OrderedDict()
    for did,k,c,v in sorted(seq):
        by.setdefault((did),{})[c]=v; by[did]['k']=k
    ids=sorted(by)
    # 6 launches per config
    for i in range(0,len(ids),6):
        grp=[by[j] for j in ids[i:i+6]]
        out={c: sum(g[c] for g in grp)/len(grp) for c in grp[0] if c!='k'}
        print(d, i//6, grp[0]['k'], {c:"%.4g"%v for c,v in out.items()})
PY
