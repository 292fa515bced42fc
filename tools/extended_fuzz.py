#!/usr/bin/env python3
"""One-off long run of the seeded random tests beyond the seeds the suite uses (tests/test_gpu_fuzz.py,
test_gpu_api_sequences.py, test_gpu_slabs.py's random chains): same checks -- the engine against the oracle or the single
domain, bit for bit -- on fresh seeds.  Prints a summary line per family; exits non-zero at the first difference.

    python tools/extended_fuzz.py [--first 200] [--count 1500]
"""
import argparse
import inspect
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=200)
    ap.add_argument("--count", type=int, default=1500)
    ap.add_argument("--seconds", type=float, default=240.0, help="stop a family after this long")
    ap.add_argument("--only", default="", help="substring of the family title to run alone")
    args = ap.parse_args()
    from oracle.oracle import Oracle
    from wayverb_amd import build
    build.build(verbose=False)
    oracle = Oracle()
    import test_gpu_fuzz as F
    families = [("random rooms / sources / receivers in seven stepping modes", F.test_random_case_equals_the_oracle_in_every_stepping_mode),
                ("speckled rooms in degenerate meshes", F.test_speckled_rooms_in_degenerate_meshes)]
    try:
        import test_gpu_api_sequences as A
        for name, fn in inspect.getmembers(A, inspect.isfunction):
            if name.startswith("test_") and "seed" in inspect.signature(fn).parameters:
                families.append(("random API sequences: " + name, fn))
    except Exception:  # noqa: BLE001
        traceback.print_exc()
    try:
        import test_gpu_slabs as S
        from wayverb_amd import engine as E

        def chains(seed, _fn=S.test_random_slab_chains_equal_the_single_domain):
            # the engine's choice of stepping, two-step passes forced with both exchanges under the march (round 4), and in round 3's
            # order (the test's _step_mode fixture)
            # ... and three-step passes forced (round 6: three exchanges per pass; slabs of fewer than six planes send the chain to two-step passes)
            for pair, early, triple in ((None, 1, None), (1, 1, None), (1, 0, None), (1, 1, 1)):
                old = dict(E.default_tuning)
                if pair is not None:
                    E.default_tuning["pair"] = pair
                    E.default_tuning["slab_early"] = early
                if triple is not None:
                    E.default_tuning.update(triple=1, tile_lists=0)
                try:
                    _fn(None, seed, "three-step-passes" if triple else ("two-step-passes" if pair else "single-steps"))
                finally:
                    E.default_tuning.clear()
                    E.default_tuning.update(old)
        families.append(("random slab chains against the single domain, single steps, two-step passes in both orders, three-step passes", chains))
    except Exception:  # noqa: BLE001
        traceback.print_exc()
    failed = False
    for title, fn in families:
        if args.only and args.only not in title:
            continue
        params = inspect.signature(fn).parameters
        t0 = time.perf_counter()
        done = 0
        for seed in range(args.first, args.first + args.count):
            kw = {}
            for p in params:
                if params[p].default is not inspect.Parameter.empty:
                    continue
                if p == "seed":
                    kw[p] = seed
                elif p == "oracle":
                    kw[p] = oracle
                elif p == "built_library":
                    kw[p] = None
                elif p == "mode":
                    kw[p] = ["default", "passes", "single-steps", "graph-replay", "graph-and-passes"][seed % 5]
                else:
                    kw[p] = None
            try:
                fn(**kw)
            except Exception:  # noqa: BLE001
                print("FAILED %s, seed %d" % (title, seed), flush=True)
                traceback.print_exc()
                failed = True
                break
            done += 1
            if time.perf_counter() - t0 > args.seconds:
                break
        print("%s: seeds %d..%d, %d passed in %.0f s" % (title, args.first, args.first + done - 1, done, time.perf_counter() - t0), flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
