#!/usr/bin/env python3
"""bench.py -- throughput of the batched sea-ice column update on MI355X (BASELINE.json metric:
column-timesteps/sec, achieved HBM GB/s vs peak).

    python bench.py --gpus N --steps K --warmup W

One bench "step" = ONE launch of the hot path that advances every column `--substeps` model time steps (a launch is
one pass of the time-loop body over the whole resident ensemble, repeated substeps times inside the kernel; columns
never communicate).  Default: 500 time steps per launch, so that the driver's 20 timed launches cover 10 000 time steps
(SURVEY.md section 8d).  Workload (config.workload): SURVEY.md section 8(d) cfg3 -- testcase 4 / SHEBA physics and
forcing tables (boundflux 2, gravity drainage, flushing, flooding, snow), `--ncol` columns per GPU, headline geometry
Nlayer 80; the initial state tiles a 256-member perturbed ensemble that was spun up 200 days from open water
(committed fixture), so the lanes of a wave hold genuinely different columns.  State resident in HBM before the timed
region.

Multi-GPU: columns are sharded by rank, no data-path collective exists (weak scaling: per-GPU columns fixed).  Under
torchrun the ranks come from RANK / LOCAL_RANK / WORLD_SIZE; a bare `python bench.py --gpus N` (N > 1) starts N fresh
rank processes itself, before this process has made any GPU call.

The JSON line also carries
  roofline      algorithmic bytes per launch / mean launch duration (HIP events on the launch stream)
  cpu_baseline  the CPU oracle (C port of the reference algorithm) timed on this host: all cores and one core
  extra         (N = 1) the same geometry started from other stages of the SHEBA year: open water, growth, the melt
                season (flushing, flooding, regridding: the unfused paths), and the first-300-days rate they imply
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
# CPU port vs the reference itself as measured ONCE in the build container (flang -O2, one Xeon core, testcase 4 from open water,
# first 3.0e6 steps: oracle 2.32e4 column-timesteps/s, reference 2.58e4).  Only quoted when the reference binary is absent: the
# cpu_baseline leg times the reference itself in the same run (cpu_reference below) and derives the factor from that.
CALIBRATION_VS_FLANG_BUILD_CONTAINER = 0.90
REF_BINARY = os.path.join(ROOT, "oracle", "_ref", "samsim_ref_dump")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_ensemble(name):
    from samsim_amd.capi import State
    z = np.load(os.path.join(GOLDEN, name))
    st = State(np.ascontiguousarray(z["lay"][:4]), np.ascontiguousarray(z["scal"]), np.ascontiguousarray(z["n_active"]))
    clock = dict(time=float(z["time"]), step=int(z["step"]), n_time_out=int(z["n_time_out"]),
                 time_counter=int(z["time_counter"]), n_outputs=int(z["n_outputs"]))
    pert = (np.ascontiguousarray(z["dT2m"]), np.ascontiguousarray(z["precip_scale"])) if "dT2m" in z.files else None
    return z, st, clock, pert


def sheba_forcing():
    f = np.load(os.path.join(GOLDEN, "sheba_forcing.npz"))
    return (f["fl_sw"], f["fl_lw"], f["T2m"], f["precip"])


def workload(args):
    """returns cfg, member states (prognostic arrays, scalars, N_active), per-member perturbation, clock, forcing, name, data"""
    from samsim_amd import testcases as tcs
    from samsim_amd.capi import State
    if args.workload == "sheba":
        z, st, clock, pert = load_ensemble(f"sheba_ensemble_{args.nlayer}.npz")
        cfg, _ = tcs.testcase4(1, nlayer=int(z["nlayer"]), n_top=int(z["n_top"]), n_bottom=int(z["n_bottom"]))
        forcing = sheba_forcing()
        name = (f"SHEBA/testcase-4 physics+forcing (cfg3), {st.ncol}-member perturbed-T2m/precip ensemble spun up 200 days "
                "from open water (tools/make_ensemble_fixture.py), members tiled over the columns")
        data = "SHEBA ERA-interim forcing tables (fixture) + synthetic per-column perturbation; synthetic spun-up ensemble (fixture)"
    elif args.workload == "cfg5":
        cfg, st, clock = tcs.config5(1, nlayer=500)
        st = State(np.ascontiguousarray(st.lay[:4]), st.scal, st.n_active)
        pert = tcs.ensemble_perturbation(4096)
        forcing = sheba_forcing()
        name = ("cfg5: Nlayer 500 (20+460+20), thick_0 4 mm, dt 2 s, synthetic saturated slab + 0.7 m snow, SHEBA forcing from "
                "day 340, gravity drainage + flooding active")
        data = ("SHEBA ERA-interim forcing tables (fixture) + synthetic per-column perturbation; synthetic saturated slab "
                "with 0.7 m of snow replicated to all columns (samsim_amd/testcases.py config5)")
    else:
        cfg, _ = tcs.testcase1(1)
        _, st, clock, _ = load_ensemble("tc1_spunup_state.npz")
        pert, forcing = None, None
        name = "testcase-1 physics (cfg2), spun-up state replicated to identical columns"
        data = "synthetic: replicated spun-up testcase-1 state (cooling plate, no forcing tables)"
    return cfg, st, pert, clock, forcing, name, data


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


def host_cores():
    """CPU cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a one-GPU box shows all 256
    hardware threads in the mask but grants a 16-core share)"""
    n = len(os.sched_getaffinity(0))
    src = "len(os.sched_getaffinity(0))"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = float(f.read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period + 0.5))
                if q < n:
                    n, src = q, f"cgroup quota {path} ({quota}/{int(period)}), affinity mask {len(os.sched_getaffinity(0))}"
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, src


def cpu_reference(nsteps):
    """The reference itself on one host core: oracle/_ref/samsim_ref_dump -- the unmodified reference physics (mo_grotz and the
    modules it calls, SAMSIM.f90:84-106 entry restated by oracle/ref_hook/ref_driver.f90) built by oracle/build_ref.sh with flang
    -O2, statically linked, writer module replaced by a hook that here writes nothing (SAMSIM_REF_QUIET) -- run as a CHILD process
    on testcase 4 from open water for `nsteps` steps in a scratch directory holding the four forcing tables, next to the port on
    exactly that run (one column, one thread).  The reference has no restart, so it cannot start from the bench's spun-up
    ensemble: the ratio of the two on the same run is what carries the port's rate on the bench workload over to the reference."""
    import shutil
    import tempfile
    from samsim_amd import testcases as tcs
    from tests.oracle_lib import oracle_solver
    if not os.path.exists(REF_BINARY):
        return None, f"{os.path.relpath(REF_BINARY, ROOT)} absent (built by oracle/build_ref.sh where /root/reference exists)"
    f = sheba_forcing()
    d = tempfile.mkdtemp(prefix="samsim_ref_")
    try:
        os.mkdir(os.path.join(d, "output"))
        for name, a in zip(("flux_sw", "flux_lw", "T2m", "precip"), f):
            np.savetxt(os.path.join(d, name + ".txt.input"), a, fmt="%.17e")      # sub_input, mo_functions.f90:304-327: list-directed reads
        env = dict(os.environ, SAMSIM_REF_QUIET="1", SAMSIM_REF_MAXSTEPS=str(nsteps), OMP_NUM_THREADS="1")
        # the child runs on a core of its own while this process times the port on another (both single-threaded); the child's
        # time is its own CPU time (user + system, os.wait4), which for a compute-bound process is its run time
        child = subprocess.Popen([REF_BINARY, "4"], cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cfg, st = tcs.testcase4(1)
        o = oracle_solver(cfg, 1)
        o.set_threads(1)
        o.set_forcing(*f)
        o.set_state(st)
        o.set_clock()
        t = time.perf_counter()
        o.step(nsteps)
        t_port = time.perf_counter() - t
        na = int(o.get_state().n_active[0])
        o.close()
        _, status, ru = os.wait4(child.pid, 0)
        child.returncode = os.waitstatus_to_exitcode(status)
        t_ref = ru.ru_utime + ru.ru_stime
        if child.returncode != 0:
            return None, f"samsim_ref_dump exited with {child.returncode}"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {"value": nsteps / t_ref, "unit": "column-timesteps/s", "cores": 1, "kind": "reference",
            "sample": f"testcase 4 (Nlayer 100) from open water, first {nsteps} steps of one column (N_active reaches {na}), "
                      f"flang -O2 build of the unmodified reference physics as a child process: {t_ref:.1f} s of CPU time; the port on "
                      f"the same run, one thread, at the same time on another core: {t_port:.1f} s",
            "seconds": t_ref, "port_on_the_same_run": {"value": nsteps / t_port, "seconds": t_port},
            "port_over_reference": (nsteps / t_port) / (nsteps / t_ref)}, None


def cpu_baseline(cfg, st, pert, clock, forcing, col0, target_s, ref_steps):
    """oracle (C port of the reference algorithm) on a bounded sample of the same workload: all cores of this process's
    affinity mask (columns over OpenMP threads), then one core"""
    from samsim_amd.capi import State
    from tests.oracle_lib import oracle_solver

    def run(threads, ncol, seconds):
        o = oracle_solver(cfg, ncol)
        o.set_threads(threads)
        if forcing is not None:
            o.set_forcing(*forcing, tile(pert[0], ncol, col0), tile(pert[1], ncol, col0))
        o.set_state(State(tile(st.lay, ncol, col0), tile(st.scal, ncol, col0), tile(st.n_active, ncol, col0).astype(np.int32)))
        o.set_clock(**clock)
        t = time.perf_counter()
        o.step(50)
        dt = time.perf_counter() - t
        nsteps = max(50, int(50 * seconds / max(dt, 1e-3)))
        t = time.perf_counter()
        o.step(nsteps)
        dt = time.perf_counter() - t
        o.close()
        return ncol * nsteps / dt, nsteps, dt

    cores, cores_source = host_cores()
    v_all, n_all, t_all = run(cores, 4 * cores, target_s)
    v_one, n_one, t_one = run(1, 4, 0.5 * target_s)
    out = {"value": v_all, "unit": "column-timesteps/s", "cores": cores, "kind": "port",
           "sample": f"{4 * cores} columns x {n_all} steps of the same workload on {cores} OpenMP threads ({t_all:.1f} s); "
                     f"one core: 4 columns x {n_one} steps ({t_one:.1f} s)",
           "cores_source": cores_source,
           "one_core": {"value": v_one, "unit": "column-timesteps/s", "cores": 1},
           "parallel_speedup_measured": v_all / v_one}
    ref, why_not = (None, "skipped (--ref-steps 0)") if ref_steps <= 0 or forcing is None else cpu_reference(ref_steps)
    if ref is not None:
        out["reference"] = ref
        out["calibration_vs_flang"] = ref["port_over_reference"]
        out["calibration_note"] = ("measured in this run, on this host: the port's rate over the reference's on testcase 4 from open "
                                   "water (cpu_baseline.reference); the reference cannot start from the bench's spun-up ensemble")
        out["reference_equivalent_on_this_workload"] = {"one_core": v_one / ref["port_over_reference"],
                                                       "all_cores": v_all / ref["port_over_reference"], "unit": "column-timesteps/s",
                                                       "how": "port on the bench workload / (port / reference on the common run)"}
    else:
        out["reference"] = {"skipped": why_not}
        out["calibration_vs_flang"] = CALIBRATION_VS_FLANG_BUILD_CONTAINER
        out["calibration_note"] = ("constant from the build container (one Xeon core, testcase 4, first 3.0e6 steps): the reference "
                                   "binary was not timed in this run")
    return out


def lib_md5():
    import samsim_amd
    path = os.environ.get("SAMSIM_HIP_LIB", samsim_amd.HIP_LIB_PATH)
    with open(path, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def source_md5():
    """md5 over the sources the library is built from (samsim_amd/csrc/*.hip|*.h|*.cpp|Makefile + include/samsim.h), in name order"""
    import glob
    h = hashlib.md5()
    files = sorted(glob.glob(os.path.join(ROOT, "samsim_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "samsim_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "samsim_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "samsim_amd", "csrc", "Makefile"),
                                                                                      os.path.join(ROOT, "include", "samsim.h")])
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


def profiled_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950
    correction applied): only quoted when the profiled library is the one loaded now -- the same binary, or (a rebuild changes the
    binary's md5 with its path) the default library built from the very sources the profiled one was built from"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None, "profiles/traffic.json missing"
    e = t.get(key)
    if not isinstance(e, dict):
        return None, f"no PMC profile committed for {key}"
    if e.get("lib_md5") == lib_md5():
        return e.get("bytes_per_launch"), f"static: {e.get('source')} (same library build, md5 {e.get('lib_md5')[:12]})"
    if "SAMSIM_HIP_LIB" not in os.environ and e.get("src_md5") and e.get("src_md5") == source_md5():
        return e.get("bytes_per_launch"), f"static: {e.get('source')} (library rebuilt from the profiled sources, source md5 {e.get('src_md5')[:12]})"
    return None, f"PMC profile {e.get('source')} was taken on another build of the library (md5 differs): not quoted"


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes, started before this process touches the
    GPU (it never does); rank 0 prints the JSON line"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # all children are watched together: when one rank dies (HIP error, missing fixture, the rendezvous port taken in the gap
    # between the probe above and the ranks' bind) its siblings would sit in the gloo barrier until its timeout -- they are ended
    # and the first failure is what this process returns
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = abs(code) or 1
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def shared_devices(ranks):
    """[(pci bus id, [ranks])] for every GPU more than one rank ran on"""
    by = {}
    for r in ranks:
        by.setdefault(r["pci_bus_id"], []).append(r["rank"])
    return [[k, v] for k, v in sorted(by.items()) if len(v) > 1]


def gather_ranks(dist, world, rank, mine):
    """every rank's record (a small dict) on rank 0, over the gloo group the barrier uses"""
    if dist is None:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out


def timed_window(g, substeps, launches, warmup, barrier):
    for _ in range(warmup):
        g.step(substeps)
    barrier()
    work_before, _ = g.get_work()
    t0 = time.perf_counter()
    # the launches are enqueued back to back and timed as one region with HIP events on the handle's streams (a step of a large
    # ensemble is two launches on two streams; no wait between steps, so the ragged end of one is filled by the next)
    device_ms = g.steps_timed(substeps, launches)
    kernel_ms = [device_ms / launches] * launches
    barrier()
    wall = time.perf_counter() - t0
    work_after, _ = g.get_work()
    nfail = int((g.get_status()[0] != 0).sum())
    return wall, kernel_ms, float(work_after - work_before), nfail


def stage_windows(g, cfg, args, forcing):
    """the same geometry started from other stages of the SHEBA year (fixtures made by tests/golden/make_stage_fixtures.py)"""
    from samsim_amd import testcases as tcs
    from samsim_amd.capi import State
    out = {}
    stages = [("day0_open_water", None)] + [(f"day{d}", f"sheba_ensemble_{args.nlayer}_day{d}.npz") for d in (75, 150, 250, 300, 345, 360)]
    if args.stages:
        stages = [st for st in stages if st[0] in args.stages.split(",")]
    for name, fixture in stages:
        if fixture is None:
            _, st0 = tcs.testcase4(1, nlayer=int(cfg.nlayer), n_top=int(cfg.n_top), n_bottom=int(cfg.n_bottom))
            st, clock = State(np.ascontiguousarray(st0.lay[:4]), st0.scal, st0.n_active), dict()
        else:
            if not os.path.exists(os.path.join(GOLDEN, fixture)):
                continue
            _, st, clock, _ = load_ensemble(fixture)
        upload_tiled(g, st, args.ncol, 0)
        g.set_clock(**clock)
        wall, kernel_ms, cells, nfail = timed_window(g, args.substeps, args.extra_launches, 1, g.synchronize)
        steps = args.extra_launches * args.substeps
        out[name] = {"column_timesteps_per_s": args.ncol * steps / wall, "layer_cell_updates_per_s": cells / wall,
                     "timesteps": steps, "mean_launch_ms": float(np.mean(kernel_ms)), "failed_columns": nfail,
                     "mean_n_active": cells / (args.ncol * steps)}
    return out


def first_300_days(rate_of_stage):
    """time-weighted rate over days 0-300 from the stage windows (each stage stands for the days around it)"""
    seg = [("day0_open_water", 0, 40), ("day75", 40, 112), ("day150", 112, 175), ("day200", 175, 225), ("day250", 225, 275),
           ("day300", 275, 300)]
    if any(s not in rate_of_stage for s, _, _ in seg):
        return None
    t = sum((b - a) / rate_of_stage[s] for s, a, b in seg)
    return 300.0 / t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=1 << 20, help="columns per GPU")
    ap.add_argument("--substeps", type=int, default=500, help="model time steps per launch")
    ap.add_argument("--workload", choices=["sheba", "tc1", "cfg5"], default="sheba")
    ap.add_argument("--nlayer", type=int, default=80,
                    help="SHEBA geometry: 80 = 20+40+20 (the N_layers BASELINE.json's metric is quoted on) or 100 = 20+60+20 "
                         "(testcase 4 as shipped)")
    ap.add_argument("--sites", type=int, default=1,
                    help="forcing sets (samsim_set_forcing_sites): the tables replicated N times, column c on set c mod N")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--ref-steps", type=int, default=1500000,
                    help="cpu_baseline: steps of testcase 4 from open water on which the reference binary and the port are timed "
                         "side by side (0 = skip; about half a minute, both at once, at the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the stage windows (open water ... melt season)")
    ap.add_argument("--extra-launches", type=int, default=2)
    ap.add_argument("--stages", default=None, help="comma-separated subset of the stage windows (day0_open_water, day75, ... day360)")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, let them meet (barrier + reductions) and report the sharding without touching a GPU")
    ap.add_argument("--device-map", default=None,
                    help="comma-separated HIP device per local rank (default: device = local rank); '0,0' rehearses two ranks "
                         "on a one-GPU box")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    device = local_rank if args.device_map is None else int(args.device_map.split(",")[local_rank])
    dist = None
    if world > 1:
        # the data path has no collective; ranks only meet for the timing barrier and the max-over-ranks reduction,
        # which run over gloo so that this process holds exactly one GPU runtime (the one the HIP library uses)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on the C-level stdout; stdout is for the ONE JSON line, so fd 1 points at stderr while
        # the ranks connect (init + a first barrier, which is when the full mesh is made)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    if args.dry_run:
        # the launch plumbing alone: every rank reports its shard, rank 0 prints what a real run would call n_gpus
        shards = [[rank, device, rank * args.ncol, args.ncol]]
        if dist is not None:
            import torch
            t = torch.zeros(world, 4, dtype=torch.int64)
            t[rank] = torch.tensor(shards[0])
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.barrier()
            shards = t.tolist()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "shards_rank_device_col0_ncol": shards,
                              "shared_devices": shared_devices([dict(rank=r, device=d, pci_bus_id=f"dry-run:{d}") for r, d, _, _ in shards])}),
                  flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    import samsim_amd

    cfg, st, pert, clock, forcing, wname, data = workload(args)
    ncol = args.ncol
    col0 = rank * ncol
    g = samsim_amd.hip_solver(cfg, ncol, device=device)
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

    wall, kernel_ms, cells, nfail = timed_window(g, args.substeps, args.steps, args.warmup, barrier)

    # who ran where: every rank reports the device its handle lives on (ordinal and PCI bus id, from the library), its column range
    # and its own rate; rank 0 prints them and refuses a record in which two ranks shared a GPU unless that was asked for
    dev_ordinal, pci = g.get_device()
    ranks = gather_ranks(dist, world, rank, {"rank": rank, "local_rank": local_rank, "device": dev_ordinal, "pci_bus_id": pci,
                                             "host": socket.gethostname(), "col0": col0, "ncol": ncol,
                                             "column_timesteps_per_s": ncol * args.steps * args.substeps / wall,
                                             "mean_launch_ms": float(np.mean(kernel_ms)), "failed_columns": nfail})
    wall_max, fails = wall, float(nfail)
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
        traffic, traffic_source = profiled_traffic(f"{args.workload}:{ncol}:{nlayer}:{args.substeps}")
        shared = shared_devices(ranks)
        if shared and args.device_map is None:
            sys.exit(f"bench.py: ranks share a GPU without --device-map: {shared} (ranks: {ranks})")
        out = {
            "metric": "column-timesteps/sec", "value": value, "unit": "column-timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": data,
            "config": {"workload": wname, "ncol_per_gpu": ncol, "nlayer": nlayer, "timesteps_per_step": args.substeps,
                       "timed_timesteps": timesteps,
                       "parallelism": f"columns sharded over {world} GPU(s), no collective", "forcing_sites": args.sites,
                       "ranks": ranks, "shared_devices": shared, "per_gpu_mean": value / world},
            "layer_cell_updates_per_s": cells / wall_max,
            "failed_columns": int(fails),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "samsim_step_kernel", "mean_launch_ms": mean_ms,
                         "how": "HIP events around the K back-to-back steps on the launch streams, / K; a step of >= 8192 column "
                                "blocks is two concurrent launches (half of the columns each, on two streams)",
                         "algorithmic_bytes_per_launch": bytes_per_colstep * ncol * args.substeps,
                         "lib_md5": lib_md5()},
        }
        if world == 1 and not args.no_extra and args.workload == "sheba" and args.sites == 1:
            ex = stage_windows(g, cfg, args, forcing)
            rates = {k: v["column_timesteps_per_s"] for k, v in ex.items()}
            rates["day200"] = value
            out["extra"] = {"stages": ex,
                            "melt_season": ex.get("day360") or ex.get("day345"),
                            "first_300_days": {"column_timesteps_per_s": first_300_days(rates),
                                               "how": "time-weighted over the stage windows day 0 / 75 / 150 / 200 (headline) / 250 / 300; "
                                                      "each stage stands for the days around it"}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, st, pert, clock, forcing, col0, args.cpu_seconds, args.ref_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
