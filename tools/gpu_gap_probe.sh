#!/bin/bash
# launch gaps of a two-step pass at 256^3 from a rocprofv3 kernel trace (with / without the engine's own event timing)
export TMPDIR=/tmp; O=gpurun_out/${1:-gaps}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-small --no-reference-on-gpu"
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o s -- $B --nx 256 --ny 256 --nz 256 --steps 200 --warmup 20 > $O/tr.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if any(k in r['Kernel_Name'] for k in ('pair_march', 'boundary_kernel'))][-240:]
agg = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = a['Kernel_Name'].split('(')[0].replace('void wv::', '')
    agg[name].append(((int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3, (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3))
for n, v in agg.items():
    print('%-40s n %3d  mean duration %.1f us  mean gap to the next launch %.1f us' % (n, len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)))
PY
rm -rf $O/tr
