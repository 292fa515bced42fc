import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np
import cases
from conftest import golden
from helpers import run_engine, set_tuning, sha
for wg in (1, 2, 8, 0):
    for name in ("impulse_flat", "random"):
        for tag in ("f32", "f64"):
            set_tuning(resident=1, pair=0, resident_workgroups=wg)
            case = cases.CASES[name]()
            r = run_engine(case, tag)
            g = golden(name)
            ok_t = np.array_equal(r["trace"].view(np.uint8), g["trace_" + tag].view(np.uint8))
            ok_c = sha(r["current"]) == str(g["sha_current_" + tag])
            ok_b = [sha(b) for b in r["bd"]] == [str(s) for s in g["sha_bd_" + tag]]
            first_bad = -1
            if not ok_t:
                bad = np.nonzero(np.any(r["trace"] != g["trace_" + tag], axis=1))[0]
                first_bad = int(bad[0])
            print("workgroups %d %s %s dims %s: trace %s (first differing step %d) current %s filters %s" % (wg, name, tag, case["mesh"].dims, ok_t, first_bad, ok_c, ok_b), flush=True)
