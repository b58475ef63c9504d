"""What sorting the permutations into tiles by an early level's null maxima would buy, measured without touching the library:
pass 1 on the masks as drawn gives every level's maxima; the masks are then re-ordered on the host by the chosen level's
maxima and a fresh plan is timed.  python tools/sorted_tiles_probe.py [method1|method2] [key level: 1b|2|3|4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GCRE_QUIET", "1")
import numpy as np
import bench
from geneticscre_amd import api

method = sys.argv[1] if len(sys.argv) > 1 else "method1"
keys = sys.argv[2:] or ["1b", "2", "3", "4"]
cfg = dict(bench.CONFIGS["roofline"]); cfg["method"] = method
prob, masks = bench.build_inputs(cfg, 20261003, 100)

def timed(m):
    plan = api.ResidentPlan(prob, packed_masks=m)
    out = plan.run()
    t0 = time.perf_counter()
    for _ in range(3):
        out = plan.run()
    t = (time.perf_counter() - t0) / 3
    prof = dict(plan.last_profile)
    plan.close()
    return t, out, prof

t, out, prof = timed(masks)
print(f"as drawn: {t * 1e3:.2f} ms/pass  null {prof['null_kernel_ms']:.2f}  lookups {prof['ie_lookup_tiles']}", flush=True)
for key in keys:
    order = np.argsort(out[key].null, kind="stable")
    t2, out2, prof2 = timed(np.ascontiguousarray(masks[order]))
    same = all(np.array_equal(out2[k].null, out[k].null[order]) for k in out)
    print(f"sorted by level {key}: {t2 * 1e3:.2f} ms/pass  null {prof2['null_kernel_ms']:.2f}  lookups {prof2['ie_lookup_tiles']}  maxima equal after un-permuting: {same}", flush=True)
