#!/bin/bash
# the race between a slab's own face push and its next source sample: does the new test catch it in the library before the fix (26eeb6f),
# and in the one before the transport's waits were narrowed (2c03078)?  does the fixed library pass, and the chain fuzz?
export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O
cp wayverb_amd/libwayverb_amd.so /tmp/fixed.so
for lib in tools/ab/libwayverb_amd_26eeb6f.so tools/ab/libwayverb_amd_2c03078.so /tmp/fixed.so; do
  cp $lib wayverb_amd/libwayverb_amd.so; touch wayverb_amd/libwayverb_amd.so wayverb_amd/csrc/engine.resources.txt
  echo "== $lib"; python -m pytest tests/test_gpu_slabs.py -q -m gpu -k "soft_source_on_a_slab_face" 2>&1 | grep "passed\|failed"
done | tee $O/own_push_race_ab.txt
timeout 400 python tools/extended_fuzz.py --first 12000 --count 6000 --seconds 240 --only "slab chains" 2>&1 | grep -v "amdgpu.ids" | tee $O/extended_fuzz_slabs_final.txt | tail -4
