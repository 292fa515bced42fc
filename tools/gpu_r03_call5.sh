#!/bin/bash
# round 3, fifth GPU call: the new long parity tests; timeline of a slab pass (8 x 128 planes); PMC bytes of the boundary launches
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_on_device.py tests/test_gpu_div3.py -x -q -m gpu -k "64_steps or 500_steps or div3" -s ) > $O/pytest_call5.txt 2>&1; grep -h "passed\|failed\|max |difference|\|real" $O/pytest_call5.txt | tail -8
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -o t -- python tools/slab_overhead.py --world 8 --steps 8 > $O/tl.log 2>&1; tail -1 $O/tl.log
python tools/pass_timeline.py $O/tl 5 2>&1 | tee $O/slab_pass_timeline.txt | head -70
rm -rf $O/tl
CMD="python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-small --no-reference-on-gpu"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o b -- $CMD > $O/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, collections, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/r03/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"].replace("void wv::", "").split("(")[0][:60]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(k, {})[c] = {"mean_KiB": sum(v) / len(v), "n": len(v)}
json.dump(out, open("gpurun_out/r03/pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    if "boundary_kernel" in k or "pair_march" in k:
        f, w = v.get("FETCH_SIZE", {}).get("mean_KiB", 0), v.get("WRITE_SIZE", {}).get("mean_KiB", 0)
        print("%-50s fetched 2 x %.0f KiB = %.3f GB, written %.3f GB, total %.3f GB" % (k, f, 2 * f * 1024 / 1e9, w * 1024 / 1e9, (2 * f + w) * 1024 / 1e9))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
