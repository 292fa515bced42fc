#!/bin/bash
# inner fix-ups (boundary entries finish the inside nodes they face) and idle filter state: tests + A/B
O=gpurun_out/x2; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-small --no-reference-on-gpu"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_ms'))" 2>/dev/null; }
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; grep -h "passed\|failed" $O/pytest.txt | tail -1
kstat() {
  python - "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('boundary_kernel', 'stream_sweep', 'pair_march', 'pair_fixup')):
        print("   %-28s calls %5s  mean %9.1f us" % (n.split('<')[0].replace('void wv::', ''), r['Calls'], float(r['AverageNs']) / 1e3))
PY
}
{
for cfg in "1 1" "0 1" "1 1" "0 1"; do
  set -- $cfg
  echo "1024^3 two-step passes, WV_PAIR_INNER_FIX=$1 WV_FLAT_SKIP=$2: $(WV_PAIR_INNER_FIX=$1 WV_FLAT_SKIP=$2 $B --steps 60 --warmup 6 | val)"
done
for cfg in "1 1" "0 0"; do
  set -- $cfg
  echo "1024^3 kernels, WV_PAIR_INNER_FIX=$1 WV_FLAT_SKIP=$2"
  WV_PAIR_INNER_FIX=$1 WV_FLAT_SKIP=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o s -- $B --steps 30 --warmup 6 > $O/tr.log 2>&1
  kstat $O/tr/s_kernel_stats.csv; rm -rf $O/tr
done
for n in 384 512 768; do for cfg in "1 1" "0 0"; do set -- $cfg
  echo "n=$n WV_PAIR_INNER_FIX=$1 WV_FLAT_SKIP=$2: $(WV_PAIR_INNER_FIX=$1 WV_FLAT_SKIP=$2 $B --nx $n --ny $n --nz $n --steps 400 --warmup 40 | val)"; done; done
for fs in 1 0; do echo "n=256 WV_FLAT_SKIP=$fs: $(WV_FLAT_SKIP=$fs $B --nx 256 --ny 256 --nz 256 --steps 3000 --warmup 50 | val)"; done
} > $O/inner_fix.txt 2>&1
cat $O/inner_fix.txt
