#!/usr/bin/env python3
"""How fast does round-off grow through freeze-up on the sites / ocean-grid path?  (CPU only; TEST INFRASTRUCTURE.)

Round 2's GPU suite once failed a FREE run of tests/test_gpu_parity.py::test_per_column_ocean_grid_of_columns at
"day 75: array H_abs rel err 4.249e-05" (64 SHEBA columns over oceans of different heat flux and salinity).  The HIP path differs
from the checker by ulp-level re-associations (shared reciprocals, Horner form, fused multiply-adds in getT's Newton step).  This
script measures what a perturbation of THAT size does to the same 64 columns with no GPU involved: the checker as built
(-ffp-contract=off) against the same source built with -march=native -ffp-contract=fast (every a*b+c contracted), both free from
open water.  Per column it records the first chunk in which any prognostic value differs by more than 1e-12 relative, and the
error at every tenth day up to day `--days`.

  python tools/freeze_up_sensitivity.py --days 100 --out profiles/r3_freeze_up_sensitivity.json
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from samsim_amd import testcases as tcs          # noqa: E402
from tests.helpers import sheba_forcing, rel_err  # noqa: E402
from tests.oracle_lib import OracleSolver, load_oracle  # noqa: E402


def build_fma():
    so = os.path.join(ROOT, "oracle", "liboracle_fma.so")
    src = os.path.join(ROOT, "oracle", "samsim_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-march=native", "-mfma", "-ffp-contract=fast", "-fno-fast-math", "-fPIC", "-fopenmp",
                               "-shared", "-o", so, src, "-lm"])
    return so


def errs(a, b):
    k = np.arange(a.nlayer)[:, None] < b.n_active[None, :]
    out = {}
    for n in ("H_abs", "S_abs", "m", "thick", "T"):
        floor = 1e-3 if n == "H_abs" else 1e-9
        x, y = a.arr(n), b.arr(n)
        e = np.where(k, np.abs(x - y) / np.maximum(np.abs(y), floor), 0.0)
        out[n] = e.max(axis=0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--days", type=int, default=100)
    ap.add_argument("--chunk", type=int, default=8641)
    ap.add_argument("--out", default=None)
    ap.add_argument("--windows", type=int, default=0, help="instead of a free run: every --every steps restart the FMA build from the "
                    "plain build's state and compare after this many steps (the protocol of the GPU test's windows)")
    ap.add_argument("--every", type=int, default=8641 * 4)
    ap.add_argument("--from-day", type=int, default=64)
    ap.add_argument("--plain-ocean", action="store_true", help="no samsim_set_ocean: the perturbed-forcing ensemble alone")
    a = ap.parse_args()
    ncol = 64
    cfg, st = tcs.testcase4(ncol)
    rng = np.random.default_rng(11)           # the ocean grid of tests/test_gpu_parity.py::test_per_column_ocean_grid_of_columns
    dq = rng.uniform(-4.0, 8.0, ncol)
    sb = rng.uniform(28.0, 36.0, ncol)
    dq[0], sb[0] = 0.0, cfg.S_bu_bottom
    dT, ps = tcs.ensemble_perturbation(ncol)
    libs = {"plain": load_oracle(), "fma": C.CDLL(build_fma())}
    sol = {}
    for name, lib in libs.items():
        s = OracleSolver(lib, "oracle_", cfg, ncol)
        s.set_threads(os.cpu_count() or 1)
        s.set_forcing(*sheba_forcing(), dT, ps)
        if not a.plain_ocean:
            s.set_ocean(dq, sb)
        s.set_state(st)
        s.set_clock()
        sol[name] = s
    if a.windows:
        o, f = sol["plain"], sol["fma"]
        o.step(8641 * a.from_day)
        recs = []
        while o.get_clock().step < 8641 * a.days:
            k = o.get_clock()
            st0 = o.get_state()
            f.set_state(st0)
            f.set_clock(time=k.time, step=k.step, n_time_out=k.n_time_out, time_counter=k.time_counter, n_outputs=k.n_outputs)
            f.step(a.windows)
            o.step(a.windows)
            sa, sb_ = f.get_state(), o.get_state()
            e = errs(sa, sb_)
            worst = np.max(np.stack(list(e.values())), axis=0)
            ts = st0.sc("thick_snow")
            rec = {"step0": int(k.step), "day0": round(k.step / 8641.0, 2), "worst_rel": float(worst.max()), "worst_col": int(worst.argmax()),
                   "cols_with_thin_snow_at_start": int(((ts > 0) & (ts < 2 * cfg.thick_min)).sum()),
                   "thick_snow_worst_col": float(ts[int(worst.argmax())]), "n_active_worst_col": int(st0.n_active[int(worst.argmax())])}
            recs.append(rec)
            print(json.dumps(rec), flush=True)
            o.step(a.every - a.windows)
        if a.out:
            with open(a.out, "w") as fh:
                json.dump({"what": "checker vs its -ffp-contract=fast build on windows restarted from the checker's state (ocean grid)",
                           "window_steps": a.windows, "every_steps": a.every, "windows": recs}, fh, indent=1)
        return 0
    first = np.full(ncol, -1, dtype=np.int64)
    na_first = np.full(ncol, -1, dtype=np.int64)
    jump = np.full(ncol, -1, dtype=np.int64)
    events = []
    days = []
    nchunks = a.days * 8641 // a.chunk
    for i in range(nchunks):
        for s in sol.values():
            s.step(a.chunk)
        sa, sb_ = sol["fma"].get_state(), sol["plain"].get_state()
        e = errs(sa, sb_)
        worst = np.max(np.stack(list(e.values())), axis=0)
        step = (i + 1) * a.chunk
        newly = (first < 0) & (worst > 1e-12)
        first[newly] = step
        for cidx in np.where((jump < 0) & (worst > 1e-9))[0]:     # what the column looked like when the difference first became large
            jump[cidx] = step
            events.append({"col": int(cidx), "step": int(step), "day": round(step / 8641.0, 3), "worst_rel": float(worst[cidx]),
                           "n_active": int(sb_.n_active[cidx]),
                           **{f"{n}_{w}": float(s_.sc(n)[cidx]) for n in ("thick_snow", "m_snow", "T_snow", "H_abs_snow", "melt_thick_snow", "T_top")
                              for w, s_ in (("plain", sb_), ("fma", sa))}})
        nad = (na_first < 0) & (sa.n_active != sb_.n_active)
        na_first[nad] = step
        day = step / 8641.0
        if (i + 1) % max(1, (10 * 8641) // a.chunk) == 0 or i + 1 == nchunks:
            rec = {"day": round(day, 2), "worst_rel": float(worst.max()), "worst_col": int(worst.argmax()),
                   "median_rel": float(np.median(worst)), "cols_above_1e-9": int((worst > 1e-9).sum()),
                   "cols_above_1e-6": int((worst > 1e-6).sum()), "n_active_differs": int((sa.n_active != sb_.n_active).sum()),
                   "by_array": {n: float(v.max()) for n, v in e.items()},
                   "thick_snow_col": float(sb_.sc("thick_snow")[int(worst.argmax())]),
                   "n_active_col": int(sb_.n_active[int(worst.argmax())])}
            days.append(rec)
            print(json.dumps(rec), flush=True)
    out = {"what": "checker (-ffp-contract=off) vs the same source built -march=native -ffp-contract=fast, 64 SHEBA columns "
                   + ("(perturbed forcing)" if a.plain_ocean else "over the ocean grid of test_per_column_ocean_grid_of_columns")
                   + ", free run from open water, CPU only",
           "chunk_steps": a.chunk, "first_step_above_1e-12": first.tolist(), "first_step_n_active_differs": na_first.tolist(),
           "first_step_above_1e-9": jump.tolist(), "state_when_first_above_1e-9": events, "thick_min": float(cfg.thick_min),
           "days": days}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
