#!/usr/bin/env python3
"""One-off GPU record for profiles/: a free-running perturbed SHEBA ensemble through the melt season and the following
freeze-up -- the regime where the fused down sweep decides per step whether the volume-fraction rows are stored (sweep_down_fused)
and the late readers (melt film, func_freeboard, flush3) rely on it (refill_psi_rows is their safety net).  Every column gets its own
T2m / precipitation perturbation (counter-based, global column id); the states tile the 256-member day-345 fixture.  Run on the
counter build of the library (-DSAMSIM_STAMPS=2) the record also holds how often the safety net ran.

    SAMSIM_HIP_LIB=samsim_amd/csrc/variants/libsamsim_hip_st2.so python tools/melt_ensemble_status.py --ncol 4096 --days 115 > profiles/r3_melt_ensemble_status.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=4096)
    ap.add_argument("--days", type=int, default=115)
    ap.add_argument("--fixture", default="sheba_ensemble_80_day345.npz")
    a = ap.parse_args()
    import samsim_amd
    from samsim_amd import testcases as tcs
    z, st, clock, _ = bench.load_ensemble(a.fixture)
    cfg, _ = tcs.testcase4(1, nlayer=int(z["nlayer"]), n_top=int(z["n_top"]), n_bottom=int(z["n_bottom"]))
    g = samsim_amd.hip_solver(cfg, a.ncol)
    dT, ps = tcs.ensemble_perturbation(a.ncol)
    g.set_forcing(*bench.sheba_forcing(), dT, ps)
    bench.upload_tiled(g, st, a.ncol, 0)
    g.set_clock(**clock)
    g.set_output_window(0, 0)
    lib = samsim_amd.load()
    counters = hasattr(lib, "samsim_debug_stamps")
    if counters:
        import ctypes as C
        buf = (C.c_ulonglong * 48)()
        assert lib.samsim_debug_stamps(buf, 1) == 0
    t0 = time.time()
    rows = []
    for d in range(a.days):
        g.step(8640)
        if d % 5 == 4 or d == a.days - 1:
            status = g.get_status()[0]
            s = g.get_state(narr=4)
            codes, counts = np.unique(status, return_counts=True)
            rows.append({"day": int(z["step"]) // 8640 + d + 1, "status": {int(c): int(n) for c, n in zip(codes, counts)},
                         "n_active_min": int(s.n_active.min()), "n_active_max": int(s.n_active.max()),
                         "snow_max_m": float(s.sc("thick_snow").max()), "columns_with_snow": int((s.sc("thick_snow") > 0).sum()),
                         "wall_s": round(time.time() - t0, 1)})
            print(rows[-1], file=sys.stderr, flush=True)
    status = g.get_status()[0]
    out = {"what": "free-running perturbed SHEBA ensemble through melt season and freeze-up (sweep_down_fused's stored-row decision)",
           "ncol": a.ncol, "nlayer": int(cfg.nlayer), "fixture": a.fixture, "days": a.days, "steps": a.days * 8640,
           "columns_stopped": int((status != 0).sum()),
           "refill_psi_rows_lane_calls": (int(buf[32]) if counters and lib.samsim_debug_stamps(buf, 0) == 0 else None),
           "flush3_lane_calls": (int(buf[29]) if counters else None), "freeboard_lane_calls": (int(buf[31]) if counters else None),
           "lib_md5": bench.lib_md5(), "rows": rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
