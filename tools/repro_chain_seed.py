#!/usr/bin/env python3
"""One seed of the suite's random slab chains (tests/test_gpu_slabs.py) many times over, in both stepping modes: how often does it
fail, and with what?  (What told a race from a deterministic bug in round 3: tools/extended_fuzz.py failed on a different seed each
run, this passed 200 times on each of them.)

    python tools/repro_chain_seed.py 9889,9696 200
"""
import sys, os, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_slabs as S
from wayverb_amd import engine as E
seeds = [int(s) for s in sys.argv[1].split(",")]
reps = int(sys.argv[2])
for seed in seeds:
    for pair in (None, 1):
        bad = {}
        for r in range(reps):
            old = dict(E.default_tuning)
            if pair is not None:
                E.default_tuning["pair"] = pair
            try:
                S.test_random_slab_chains_equal_the_single_domain(None, seed, "two-step-passes" if pair else "single-steps")
            except AssertionError as e:
                bad[str(e)[:60]] = bad.get(str(e)[:60], 0) + 1
            finally:
                E.default_tuning.clear(); E.default_tuning.update(old)
        print("seed %d mode %s: %d runs, failures %s" % (seed, "passes" if pair else "default", reps, bad), flush=True)
