#!/usr/bin/env python3
"""TEST INFRASTRUCTURE / fixture generator: perturbed SHEBA ensemble (testcase-4 physics, Nlayer 80 = 20+40+20, the same
counter-based T2m / precipitation perturbation as bench.py) integrated from open water with the CPU oracle, with the
prognostic state of every member saved at chosen days:

    tests/golden/sheba_ensemble_80_day<D>.npz     (same keys as sheba_ensemble_80.npz)

bench.py times windows of the hot path started from these stages (growth from open water ... melt season) next to the
day-200 headline window.  Run in the build container:  python tests/golden/make_stage_fixtures.py [--threads 6]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from samsim_amd import testcases as tcs  # noqa: E402
from tests.oracle_lib import oracle_solver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=256)
    ap.add_argument("--nlayer", type=int, default=80)
    ap.add_argument("--days", type=int, nargs="+", default=[25, 75, 150, 250, 300, 345, 360])
    ap.add_argument("--threads", type=int, default=6)
    a = ap.parse_args()
    n_tb = 20
    cfg, st = tcs.testcase4(a.members, nlayer=a.nlayer, n_top=n_tb, n_bottom=n_tb)
    z = np.load(os.path.join(ROOT, "tests", "golden", "sheba_forcing.npz"))
    dT, ps = tcs.ensemble_perturbation(a.members)
    o = oracle_solver(cfg, a.members)
    o.set_threads(a.threads)
    o.set_forcing(z["fl_sw"], z["fl_lw"], z["T2m"], z["precip"], dT, ps)
    o.set_state(st)
    o.set_clock()
    t0 = time.time()
    day = 0
    for target in sorted(a.days):
        while day < target:
            o.step(8640)
            day += 1
        s = o.get_state()
        status = o.get_status()[0]
        clk = o.get_clock()
        ok = status == 0
        out = os.path.join(ROOT, "tests", "golden", f"sheba_ensemble_{a.nlayer}_day{target}.npz")
        np.savez_compressed(out, lay=s.lay[:4][:, :, ok], scal=s.scal[:, ok], n_active=s.n_active[ok], dT2m=dT[ok],
                            precip_scale=ps[ok], time=clk.time, step=clk.step, n_time_out=clk.n_time_out,
                            time_counter=clk.time_counter, n_outputs=clk.n_outputs, nlayer=a.nlayer, n_top=n_tb, n_bottom=n_tb)
        print(f"day {target}: {time.time() - t0:.0f} s, failed {int((~ok).sum())}, N_active {s.n_active.min()}..{s.n_active.max()}, "
              f"snow {s.sc('thick_snow').min():.3f}..{s.sc('thick_snow').max():.3f} m", flush=True)


if __name__ == "__main__":
    main()
