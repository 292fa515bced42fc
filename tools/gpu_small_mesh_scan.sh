#!/bin/bash
# where do two-step passes start to pay?  (pair_min_nodes_ in engine.hip)  + chunking of the march on small meshes
O=gpurun_out/${1:-scan}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; grep -h "passed\|failed" $O/pytest.txt | tail -1
B="python bench.py --no-cpu-baseline --no-small --no-reference-on-gpu"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms'))" 2>/dev/null; }
kstat() {
  python - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('boundary_kernel', 'stream_sweep', 'pair_march', 'pair_fixup', 'pre_post')):
        print("   %-28s calls %5s  mean %9.1f us" % (n.split('<')[0].replace('void wv::', ''), r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
{
for n in 64 96 128 160 192 224 256 288 320 352; do for p in 0 1; do
  echo "n=$n WV_PAIR=$p: $($B --tuning pair=$p --nx $n --ny $n --nz $n --steps 3000 --warmup 100 | val)"; done; done
for c in 8 16 32; do echo "n=256 WV_PAIR=1 WV_PAIR_CHUNKS=$c: $($B --tuning pair=1,pair_chunks=$c --nx 256 --ny 256 --nz 256 --steps 3000 --warmup 100 | val)"; done
for p in 1 0; do
echo "256^3 kernels WV_PAIR=$p"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o s -- $B --tuning pair=$p --nx 256 --ny 256 --nz 256 --steps 600 --warmup 20 > $O/tr.log 2>&1
kstat $O/tr/s_kernel_stats.csv; rm -rf $O/tr
done
} > $O/mid.txt 2>&1
cat $O/mid.txt
echo "1024^3: $(python bench.py --no-cpu-baseline --no-small --no-reference-on-gpu --steps 60 --warmup 6 | val)" | tee -a $O/mid.txt
