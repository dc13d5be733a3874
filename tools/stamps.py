#!/usr/bin/env python3
"""Region timing / event counters of samsim_step_kernel on the bench workload (profiling builds of the library only:
-DSAMSIM_STAMPS=1 for the s_memtime regions, =2 for the counters; select the build with SAMSIM_HIP_LIB=...).

    SAMSIM_HIP_LIB=samsim_amd/csrc/variants/libsamsim_hip_st1.so python3 tools/stamps.py [--nlayer 80] [--launches 3]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = ["t_prologue", "t_down_fused", "t_down_unfused", "t_surface", "t_up", "t_post", "t_head", "t_tail",
         "wave_steps", "fused", "unfused", "up_trips", "newton_wave", "newton_lane", "lanes", "down_trips", "drain_wave",
         "drain_lane", "dirty", "lanes_coupling", "t_up_head", "t_up_getT", "t_up_tail", "t_down_A", "t_down_BC",
         "lanes_flood_possible", "lanes_irregular", "lanes_dirty", "lanes_unfused", "lanes_flush3", "lanes_regrid", "lanes_freeboard",
         "lanes_refill_psi", "down_rows_interior", "down_rows_without_expulsion", "getT_lanes_redone_by_the_general_routine",
         "getT_waves_with_such_a_lane", "their_evaluations_wave_max", "up_sweeps_without_the_next_first_sweep"] + [f"slot{i}" for i in range(39, 48)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=1 << 20)
    ap.add_argument("--nlayer", type=int, default=80)
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--workload", default="sheba")
    ap.add_argument("--sites", type=int, default=1)
    ap.add_argument("--fixture", default=None, help="stage fixture to start from (e.g. sheba_ensemble_80_day360.npz)")
    args = ap.parse_args()
    import samsim_amd
    cfg, st, pert, clock, forcing, wname, _ = bench.workload(args)
    if args.fixture:
        _, st, clock, _ = bench.load_ensemble(args.fixture)
    g = samsim_amd.hip_solver(cfg, args.ncol, device=0)
    if forcing is not None:
        g.set_forcing(*forcing, bench.tile(pert[0], args.ncol, 0), bench.tile(pert[1], args.ncol, 0))
    bench.upload_tiled(g, st, args.ncol, 0)
    g.set_clock(**clock)
    g.set_output_window(0, 0)
    lib = samsim_amd.load()
    buf = (C.c_ulonglong * 48)()
    g.step(args.substeps)
    g.synchronize()
    assert lib.samsim_debug_stamps(buf, 1) == 0
    ms = [g.step_timed(args.substeps) for _ in range(args.launches)]
    g.synchronize()
    assert lib.samsim_debug_stamps(buf, 0) == 0
    v = np.array(list(buf), dtype=np.float64)
    out = {"lib": os.environ.get("SAMSIM_HIP_LIB", "default"), "mean_launch_ms": float(np.mean(ms)),
           "launches": args.launches, "nlayer": args.nlayer}
    out.update({n: v[i] for i, n in enumerate(NAMES)})
    tidx = list(range(8)) + list(range(20, 25))
    tsum = v[tidx].sum()
    if tsum > 0:
        out["share"] = {NAMES[i]: round(v[i] / tsum, 4) for i in tidx}
    if v[11] > 0:
        out["newton_evals_per_cell_wave_max"] = v[12] / v[11]
        out["newton_evals_per_cell_lane_mean"] = v[13] / (v[11] * 64)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
