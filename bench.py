#!/usr/bin/env python3
"""bench.py -- throughput of the batched sea-ice column update on MI355X (BASELINE.json metric:
column-timesteps/sec, achieved HBM GB/s vs peak).

    python bench.py --gpus N --steps K --warmup W

One bench "step" = ONE launch of the hot path that advances every column `--substeps` model time steps
(a launch is one pass of the time-loop body over the whole resident ensemble, repeated substeps times inside
the kernel; columns never communicate).  Workload (config.workload): SURVEY.md section 8(d) cfg3 -- testcase 4 /
SHEBA physics and forcing tables (boundflux 2, gravity drainage, flushing, flooding, snow), `--ncol` columns per
GPU; the initial state tiles a 256-member perturbed ensemble that was spun up 200 days from open water (committed
fixture, tools/make_ensemble_fixture.py), so the lanes of a wave hold genuinely different columns.  The state is resident in HBM before the timed region.
Multi-GPU: columns are sharded by rank, no data-path collective exists (weak scaling: per-GPU columns fixed).

The JSON line also carries
  roofline      algorithmic bytes per launch / mean launch duration (HIP events on the launch stream)
  cpu_baseline  the CPU oracle (C port of the reference algorithm) timed on this host on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def load_checkpoint(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    from samsim_amd.capi import State
    st = State(np.ascontiguousarray(z["lay"]), np.ascontiguousarray(z["scal"]), np.ascontiguousarray(z["n_active"]))
    clock = dict(time=float(z["time"]), step=int(z["step"]), n_time_out=int(z["n_time_out"]),
                 time_counter=int(z["time_counter"]), n_outputs=int(z["n_outputs"]))
    return st, clock


def workload(args):
    """returns cfg, member states (prognostic arrays, scalars, N_active), per-member perturbation, clock, forcing, name"""
    from samsim_amd import testcases as tcs
    from samsim_amd.capi import State
    if args.workload == "sheba":
        z = np.load(os.path.join(ROOT, "tests", "golden", f"sheba_ensemble_{args.nlayer}.npz"))
        cfg, _ = tcs.testcase4(1, nlayer=int(z["nlayer"]), n_top=int(z["n_top"]), n_bottom=int(z["n_bottom"]))
        st = State(np.ascontiguousarray(z["lay"]), np.ascontiguousarray(z["scal"]), np.ascontiguousarray(z["n_active"]))
        pert = (np.ascontiguousarray(z["dT2m"]), np.ascontiguousarray(z["precip_scale"]))
        clock = dict(time=float(z["time"]), step=int(z["step"]), n_time_out=int(z["n_time_out"]),
                     time_counter=int(z["time_counter"]), n_outputs=int(z["n_outputs"]))
        f = np.load(os.path.join(ROOT, "tests", "golden", "sheba_forcing.npz"))
        forcing = (f["fl_sw"], f["fl_lw"], f["T2m"], f["precip"])
        name = (f"SHEBA/testcase-4 physics+forcing (cfg3), {st.ncol}-member perturbed-T2m/precip ensemble spun up 200 days "
                "from open water (tools/make_ensemble_fixture.py), members tiled over the columns")
    elif args.workload == "cfg5":
        cfg, st, clock = tcs.config5(1, nlayer=500)
        st = State(np.ascontiguousarray(st.lay[:4]), st.scal, st.n_active)
        pert = tcs.ensemble_perturbation(4096)
        f = np.load(os.path.join(ROOT, "tests", "golden", "sheba_forcing.npz"))
        forcing = (f["fl_sw"], f["fl_lw"], f["T2m"], f["precip"])
        name = ("cfg5: Nlayer 500 (20+460+20), thick_0 4 mm, dt 2 s, synthetic saturated slab + 0.7 m snow, SHEBA forcing from "
                "day 340, gravity drainage + flooding active")
    else:
        cfg, _ = tcs.testcase1(1)
        st, clock = load_checkpoint("tc1_spunup_state.npz")
        st = State(np.ascontiguousarray(st.lay[:4]), st.scal, st.n_active)
        pert, forcing = None, None
        name = "testcase-1 physics (cfg2), spun-up state replicated to identical columns"
    return cfg, st, pert, clock, forcing, name


def tile(a, n, col0=0):
    """member = global column id mod nmember, along the last axis"""
    idx = (np.arange(col0, col0 + n) % a.shape[-1])
    return np.ascontiguousarray(a[..., idx])


def upload_tiled(solver, st, ncol, col0, chunk=65536):
    """prognostic arrays + scalars of member (global column id mod nmember), uploaded in chunks"""
    from samsim_amd.capi import State
    c0 = 0
    while c0 < ncol:
        n = min(chunk, ncol - c0)
        solver.set_state(State(tile(st.lay, n, col0 + c0), tile(st.scal, n, col0 + c0),
                               tile(st.n_active, n, col0 + c0).astype(np.int32)), c0)
        c0 += n


def cpu_baseline(cfg, st, pert, clock, forcing, col0, target_s):
    """oracle (C port of the reference algorithm) on a bounded sample of the same workload"""
    from samsim_amd.capi import State
    from tests.oracle_lib import oracle_solver
    # a one-GPU box grants a 16-core CPU share whatever the affinity mask says
    cores = min(16, len(os.sched_getaffinity(0)))
    ncol = 4 * cores
    o = oracle_solver(cfg, ncol)
    o.set_threads(cores)
    if forcing is not None:
        o.set_forcing(*forcing, tile(pert[0], ncol, col0), tile(pert[1], ncol, col0))
    o.set_state(State(tile(st.lay, ncol, col0), tile(st.scal, ncol, col0), tile(st.n_active, ncol, col0).astype(np.int32)))
    o.set_clock(**clock)
    t = time.perf_counter()
    o.step(50)
    dt = time.perf_counter() - t
    nsteps = max(50, int(50 * target_s / max(dt, 1e-3)))
    t = time.perf_counter()
    o.step(nsteps)
    dt = time.perf_counter() - t
    return {"value": ncol * nsteps / dt, "unit": "column-timesteps/s", "cores": cores, "kind": "port",
            "sample": f"{ncol} columns x {nsteps} steps of the same workload on {cores} OpenMP threads ({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=1 << 20, help="columns per GPU")
    ap.add_argument("--substeps", type=int, default=20, help="model time steps per launch")
    ap.add_argument("--workload", choices=["sheba", "tc1", "cfg5"], default="sheba")
    ap.add_argument("--nlayer", type=int, default=80,
                    help="SHEBA geometry: 80 = 20+40+20 (the N_layers BASELINE.json's metric is quoted on) or 100 = 20+60+20 "
                         "(testcase 4 as shipped)")
    ap.add_argument("--sites", type=int, default=1,
                    help="forcing sets (samsim_set_forcing_sites): the tables replicated N times, column c on set c mod N")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # the data path has no collective; ranks only meet for the timing barrier and the max-over-ranks reduction,
        # which run over gloo so that this process holds exactly one GPU runtime (the one the HIP library uses)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import samsim_amd
    from samsim_amd import testcases as tcs

    cfg, st, pert, clock, forcing, wname = workload(args)
    ncol = args.ncol
    col0 = rank * ncol
    g = samsim_amd.hip_solver(cfg, ncol, device=local_rank)
    if forcing is not None and args.sites > 1:
        g.set_forcing_sites(*[np.tile(a, (args.sites, 1)) for a in forcing], (np.arange(ncol) % args.sites).astype(np.int32),
                            tile(pert[0], ncol, col0), tile(pert[1], ncol, col0))
    elif forcing is not None:
        g.set_forcing(*forcing, tile(pert[0], ncol, col0), tile(pert[1], ncol, col0))
    upload_tiled(g, st, ncol, col0)
    g.set_clock(**clock)
    g.set_output_window(0, 0)

    def barrier():
        g.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        g.step(args.substeps)
    barrier()
    work_before, _ = g.get_work()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(g.step_timed(args.substeps))
    barrier()
    wall = time.perf_counter() - t0
    work_after, _ = g.get_work()
    status = g.get_status()[0]
    nfail = int((status != 0).sum())

    wall_max, cells, fails = wall, float(work_after - work_before), float(nfail)
    if dist is not None:
        import torch
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_max = float(t[0])
        s = torch.tensor([cells, fails], dtype=torch.float64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        cells, fails = float(s[0]), float(s[1])

    if rank == 0:
        timesteps = args.steps * args.substeps
        value = world * ncol * timesteps / wall_max
        nlayer = int(cfg.nlayer)
        bytes_per_colstep = 16.0 * (4 * nlayer + 24)   # read + write once of the prognostic state
        mean_ms = float(np.mean(kernel_ms))
        achieved = bytes_per_colstep * ncol * args.substeps / (mean_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE
        # in separate passes, gfx950 correction applied: profiles/r1_hbm_counter_calibration.txt); None if this
        # configuration has not been profiled
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f).get(f"{args.workload}:{ncol}:{nlayer}:{args.substeps}")
        except OSError:
            pass
        out = {
            "metric": "column-timesteps/sec", "value": value, "unit": "column-timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "SHEBA ERA-interim forcing tables (fixture) + synthetic per-column perturbation; synthetic spun-up ensemble (fixture)"
                    if args.workload == "sheba" else "synthetic: replicated spun-up testcase-1 state",
            "config": {"workload": wname, "ncol_per_gpu": ncol, "nlayer": nlayer, "timesteps_per_step": args.substeps,
                       "parallelism": f"columns sharded over {world} GPU(s), no collective", "forcing_sites": args.sites},
            "layer_cell_updates_per_s": cells / wall_max,
            "failed_columns": int(fails),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "samsim_step_kernel", "mean_launch_ms": mean_ms,
                         "algorithmic_bytes_per_launch": bytes_per_colstep * ncol * args.substeps},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, st, pert, clock, forcing, col0, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
