#!/usr/bin/env python3
"""Spins up a perturbed SHEBA ensemble ON THE GPU (HIP path) from open water and stores the prognostic state of its
members as a bench fixture: tests/golden/sheba_ensemble_<nlayer>.npz.  bench.py tiles the members over the columns of
the run (member = column mod nmember, with that member's perturbation), so that the lanes of a wave hold genuinely
different columns (different snow depth, ice thickness, regrid / melt timing) instead of replicas.

    gpurun -- python tools/make_ensemble_fixture.py --days 200 --nlayer 100 --out gpurun_out/sheba_ensemble_100.npz
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import samsim_amd  # noqa: E402
from samsim_amd import testcases as tcs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=256)
    ap.add_argument("--days", type=int, default=200)
    ap.add_argument("--nlayer", type=int, default=100)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    n_tb = 20
    cfg, st = tcs.testcase4(a.members, nlayer=a.nlayer, n_top=n_tb, n_bottom=n_tb)
    z = np.load(os.path.join(ROOT, "tests", "golden", "sheba_forcing.npz"))
    dT, ps = tcs.ensemble_perturbation(a.members)
    g = samsim_amd.hip_solver(cfg, a.members)
    g.set_forcing(z["fl_sw"], z["fl_lw"], z["T2m"], z["precip"], dT, ps)
    g.set_state(st)
    g.set_clock()
    t = time.time()
    for d in range(a.days):
        g.step(8640)
        if d % 20 == 19:
            g.synchronize()
            print(f"day {d + 1}: {time.time() - t:.0f} s", flush=True)
    s = g.get_state()
    status = g.get_status()[0]
    clk = g.get_clock()
    print("failed members:", int((status != 0).sum()), "N_active min/max:", s.n_active.min(), s.n_active.max(),
          "snow min/max:", s.sc("thick_snow").min(), s.sc("thick_snow").max())
    ok = status == 0
    np.savez_compressed(a.out, lay=s.lay[:4][:, :, ok], scal=s.scal[:, ok], n_active=s.n_active[ok], dT2m=dT[ok],
                        precip_scale=ps[ok], time=clk.time, step=clk.step, n_time_out=clk.n_time_out,
                        time_counter=clk.time_counter, n_outputs=clk.n_outputs, nlayer=a.nlayer, n_top=n_tb, n_bottom=n_tb)


if __name__ == "__main__":
    main()
