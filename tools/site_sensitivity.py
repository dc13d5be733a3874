#!/usr/bin/env python3
"""Round-off sensitivity of free runs on the five later ERA-interim sites (CPU only; TEST INFRASTRUCTURE): the checker against its own
-ffp-contract=fast build (tools/freeze_up_sensitivity.py builds it), testcase 4, one unperturbed column per site, free from open
water for 150 output days.  Tells which site's free run meets an amplifying freeze-up event, i.e. up to which day a free-running
GPU column can be held to the parity bar against the reference's records (tests/test_gpu_secondary.py).

  python tools/site_sensitivity.py > profiles/r3_site_sensitivity.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from samsim_amd import testcases as tcs                      # noqa: E402
from tests.helpers import golden                              # noqa: E402
from tests.oracle_lib import OracleSolver, load_oracle        # noqa: E402
from tools.freeze_up_sensitivity import build_fma, errs       # noqa: E402

SITES = ["75N180E", "80N00E", "75N00W", "85N180E", "80N90E"]


def main():
    zm = golden("era_sites_forcing_more.npz")
    tables = [np.stack([zm[f"{s}_{n}"] for s in SITES]) for n in ("fl_sw", "fl_lw", "T2m", "precip")]
    cfg, st = tcs.testcase4(len(SITES))
    sol = {}
    for name, lib in (("plain", load_oracle()), ("fma", C.CDLL(build_fma()))):
        s = OracleSolver(lib, "oracle_", cfg, len(SITES))
        s.set_threads(len(SITES))
        s.set_forcing_sites(*tables, np.arange(len(SITES), dtype=np.int32), None, None)
        s.set_state(st)
        s.set_clock()
        sol[name] = s
    first = {s: None for s in SITES}
    rows = []
    for d in range(150):
        for s in sol.values():
            s.step(8641)
        a, b = sol["fma"].get_state(), sol["plain"].get_state()
        w = np.max(np.stack(list(errs(a, b).values())), axis=0)
        for i, s in enumerate(SITES):
            if first[s] is None and w[i] > 1e-9:
                first[s] = d + 1
        if d % 10 == 9 or d in (59, 60, 61):
            rows.append({"day": d + 1, "worst_rel": {s: float(w[i]) for i, s in enumerate(SITES)}, "n_active": b.n_active.tolist()})
            print(rows[-1], file=sys.stderr, flush=True)
    print(json.dumps({"what": "checker vs its -ffp-contract=fast build, testcase 4 free from open water on the five later ERA-interim "
                              "sites, one column each, CPU only", "first_output_day_above_1e-9": first, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
