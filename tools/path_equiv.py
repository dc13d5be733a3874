#!/usr/bin/env python3
"""Do the fused and the unfused order of a time step give a column the same bits?  Runs a perturbed SHEBA ensemble from a stage
fixture for --steps steps with the library selected by SAMSIM_HIP_LIB and writes the prognostic state; run it once per build
(-DSAMSIM_PATH_MODE=0 per column, =1 always unfused, =2 per wave) and compare the files with --compare.

    SAMSIM_HIP_LIB=.../libsamsim_hip_pm0.so python tools/path_equiv.py --out gpurun_out/pm0.npz
    SAMSIM_HIP_LIB=.../libsamsim_hip_pm1.so python tools/path_equiv.py --out gpurun_out/pm1.npz
    python tools/path_equiv.py --compare gpurun_out/pm0.npz gpurun_out/pm1.npz
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--fixture", default="sheba_ensemble_80_day345.npz")
    ap.add_argument("--cfg5", action="store_true", help="the Nlayer-500 flooding slab of BASELINE cfg5 (every column floods in every step)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--compare", nargs=2, default=None)
    a = ap.parse_args()
    if a.compare:
        x, y = np.load(a.compare[0]), np.load(a.compare[1])
        names = ["H_abs", "S_abs", "m", "thick", "T", "phi", "psi_s", "psi_l", "psi_g", "S_bu", "S_br", "ray", "perm", "flush_v", "flush_h"]
        must = {"H_abs", "S_abs", "m", "thick", "T", "phi", "psi_s", "psi_l", "psi_g", "S_bu", "flush_v", "flush_h", "scal", "n_active", "status"}
        out, ok = {}, True

        def cmp(k, u, v):
            nonlocal ok
            same = np.array_equal(u, v, equal_nan=True)
            out[k] = "bitwise equal" if same else f"DIFFER in {int((u != v).sum())} of {u.size}: max abs {float(np.nanmax(np.abs(u.astype(float) - v.astype(float))))}"
            if not same and k in must:
                ok = False
        for k in ("scal", "n_active", "status"):
            cmp(k, x[k], y[k])
        # active layers only (rows below N_active hold what the last regrid left, which both orders leave alone)
        na = x["n_active"]
        active = np.arange(x["lay"].shape[1])[:, None] < na[None, :]
        for i, n in enumerate(names[:x["lay"].shape[0]]):
            cmp(n, np.where(active, x["lay"][i], 0.0), np.where(active, y["lay"][i], 0.0))
        out["required"] = sorted(must)
        out["note"] = "S_br, perm and ray are work arrays: S_br is written by the unfused order only, ray rows are stored where a wave drains"
        out["verdict"] = "same bits" if ok else "DIFFERENT"
        print(json.dumps(out, indent=1))
        sys.exit(0 if ok else 1)
    import samsim_amd
    from samsim_amd import testcases as tcs
    if a.cfg5:
        from samsim_amd.capi import State
        cfg, st, clock = tcs.config5(1, nlayer=500)
        st = State(np.ascontiguousarray(st.lay[:4]), st.scal, st.n_active)
    else:
        z, st, clock, _ = bench.load_ensemble(a.fixture)
        cfg, _ = tcs.testcase4(1, nlayer=int(z["nlayer"]), n_top=int(z["n_top"]), n_bottom=int(z["n_bottom"]))
    g = samsim_amd.hip_solver(cfg, a.ncol)
    dT, ps = tcs.ensemble_perturbation(a.ncol)
    g.set_forcing(*bench.sheba_forcing(), dT, ps)
    bench.upload_tiled(g, st, a.ncol, 0)
    g.set_clock(**clock)
    g.set_output_window(0, 0)
    g.step(a.steps)
    s = g.get_state()
    status = g.get_status()[0]
    np.savez(a.out, lay=s.lay, scal=s.scal, n_active=s.n_active, status=status)
    print(json.dumps({"lib": os.environ.get("SAMSIM_HIP_LIB", "default"), "steps": a.steps, "ncol": a.ncol,
                      "stopped": int((status != 0).sum()), "n_active_min": int(s.n_active.min()), "n_active_max": int(s.n_active.max())}))


if __name__ == "__main__":
    main()
