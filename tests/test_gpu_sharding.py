"""BASELINE cfg4 on the hardware at hand: the perturbed SHEBA ensemble split into contiguous column ranges, one fresh process
per range, each driving libsamsim_hip on its own handle (no collective, no exchange).  On the one-GPU box
both ranks use device 0; on an 8-GPU node bench.py maps rank -> device.  The concatenation of the shards must equal the
single-handle run bit for bit -- through the Python mirror, through bench.py's own launcher, and through the Fortran host
(col0 / ncol_total of &samsim_run)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from samsim_amd.shard import shard_range
from tests.helpers import ROOT

pytestmark = pytest.mark.gpu

NCOL, NSTEPS = 4096 + 192, 300   # not a multiple of the ranks x wave size: ragged shards


def test_two_hip_shards_on_one_device_equal_the_single_handle_run(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), "2", str(NCOL), str(NSTEPS),
                               "0", str(tmp_path / f"shard{r}.npz")], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    from tests.shard_worker import ensemble_shard
    whole, status = ensemble_shard(0, NCOL, NSTEPS)
    assert not status.any()
    covered = 0
    for r in range(2):
        z = np.load(tmp_path / f"shard{r}.npz")
        c0, n = shard_range(NCOL, r, 2)
        assert int(z["col0"]) == c0 and z["lay"].shape[2] == n
        assert np.array_equal(z["lay"][:9], whole.lay[:9, :, c0:c0 + n])     # prognostic arrays + T, phi, psi_s, psi_l, psi_g
        assert np.array_equal(z["scal"], whole.scal[:, c0:c0 + n])
        assert np.array_equal(z["n_active"], whole.n_active[c0:c0 + n])
        assert not z["status"].any()
        covered += n
    assert covered == NCOL
    # the shards really are different columns (the perturbation follows the global index)
    assert len(np.unique(whole.sc("m_snow"))) > NCOL // 2


def test_bench_launcher_runs_two_ranks_on_the_gpu():
    """`python bench.py --gpus 2` without torchrun: the parent starts two fresh rank processes (here both on device 0), the
    ranks shard the columns, meet for the timing barrier, and rank 0 reports n_gpus = 2 with the work of both"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device-map", "0,0", "--ncol", "65536",
                          "--substeps", "20", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["failed_columns"] == 0 and d["scaling"] == "weak"
    assert d["value"] == pytest.approx(2 * 65536 * 40 / (d["ms_per_step"] * 2e-3), rel=1e-6)
    # both ranks' layer-cell updates are in the sum: about 80 layers x 2 x 65536 columns x 40 steps
    assert d["layer_cell_updates_per_s"] * d["ms_per_step"] * 2e-3 > 1.5 * 65536 * 40 * 70


def test_fortran_host_column_ranges(tmp_path):
    """two host processes with col0 = 0 / 96 and ncol = 96 reproduce the columns of one host process with ncol = 192: same
    restart files column for column (testcase 4 with the per-column perturbation, 200 steps)"""
    from tests.test_gpu_fortran_host import HOST, run_host
    if not os.path.exists(HOST):
        pytest.skip("Fortran host not built (no flang)")
    from samsim_amd import checkpoint

    def run(name, ncol, col0):
        d = tmp_path / name
        d.mkdir()
        out = run_host(d, f"&samsim_run\n testcase = 4, ncol = {ncol}, col0 = {col0}, ncol_total = 192, perturb = .true., "
                          f"max_steps = 200,\n restart_out = 'state.chk' /\n")
        assert "restart file written" in out
        return d / "state.chk"

    def read(path):
        h = checkpoint.read_header(str(path))
        import struct
        with open(path, "rb") as f:
            f.seek(checkpoint._HDR.size)
            c0, n = struct.unpack("<2q", f.read(16))
            lay = np.frombuffer(f.read(8 * h["narr"] * h["nlayer"] * n), dtype="<f8").reshape(h["narr"], h["nlayer"], n)
            scal = np.frombuffer(f.read(8 * h["nscal"] * n), dtype="<f8").reshape(h["nscal"], n)
            na = np.frombuffer(f.read(4 * n), dtype="<i4")
        assert (c0, n) == (0, h["ncol"])
        return lay, scal, na

    whole = read(run("whole", 192, 0))
    for name, c0 in (("a", 0), ("b", 96)):
        lay, scal, na = read(run(name, 96, c0))
        assert np.array_equal(lay[:9], whole[0][:9, :, c0:c0 + 96])
        assert np.array_equal(scal, whole[1][:, c0:c0 + 96])
        assert np.array_equal(na, whole[2][c0:c0 + 96])
    from samsim_amd.capi import S
    assert len(np.unique(whole[1][S["T2m"]])) > 96     # the perturbed air temperature differs from column to column


def test_a_column_does_not_depend_on_its_wave_mates():
    """The kernel takes some decisions per wave (one order of the step for all 64 columns when any of them has thin snow or a
    flooded surface; Rayleigh-number rows stored where any
    of them drains).  None of them may change a column's bits: the same 256 melt-season columns (day 345: snow-covered, thin
    snow, bare, flushing) are run in their order and in a shuffled order -- other wave-mates, other decisions -- and every
    column must come out identical."""
    import bench
    import samsim_amd
    from samsim_amd import testcases as tcs
    from samsim_amd.capi import State
    z, st, clock, pert = bench.load_ensemble("sheba_ensemble_80_day345.npz")
    cfg, _ = tcs.testcase4(1, nlayer=int(z["nlayer"]), n_top=int(z["n_top"]), n_bottom=int(z["n_bottom"]))
    n = st.ncol
    rng = np.random.default_rng(7)
    results = []
    for perm in (np.arange(n), rng.permutation(n)):
        g = samsim_amd.hip_solver(cfg, n)
        g.set_forcing(*bench.sheba_forcing(), np.ascontiguousarray(pert[0][perm]), np.ascontiguousarray(pert[1][perm]))
        g.set_state(State(np.ascontiguousarray(st.lay[..., perm]), np.ascontiguousarray(st.scal[..., perm]),
                          np.ascontiguousarray(st.n_active[perm]).astype(np.int32)))
        g.set_clock(**clock)
        g.set_output_window(0, 0)
        g.step(1500)
        s, status = g.get_state(), g.get_status()[0]
        inv = np.argsort(perm)
        results.append((s.lay[..., inv], s.scal[..., inv], s.n_active[inv], status[inv]))
        g.close()
    (la, sa, na, xa), (lb, sb, nb, xb) = results
    assert not xa.any() and np.array_equal(xa, xb) and np.array_equal(na, nb)
    assert np.array_equal(sa, sb)
    act = np.arange(la.shape[1])[:, None] < na[None, :]
    for i, name in enumerate(["H_abs", "S_abs", "m", "thick", "T", "phi", "psi_s", "psi_l", "psi_g", "S_bu"]):
        assert np.array_equal(np.where(act, la[i], 0.0), np.where(act, lb[i], 0.0)), name
    # the ensemble really is in the regime the decisions are about: part of it under thin snow, part of it bare
    thin = (sa[bench_scalar_index("thick_snow")] > 0) & (sa[bench_scalar_index("thick_snow")] < float(cfg.thick_min))
    assert 0 < thin.sum() < n or (sa[bench_scalar_index("thick_snow")] == 0).any()


def bench_scalar_index(name):
    from samsim_amd.capi import SCALARS
    return list(SCALARS).index(name)


def test_a_step_split_over_two_streams_equals_the_single_launch():
    """A step of a large ensemble (>= 8 192 column blocks) is two launches on two streams, half of the blocks each; here the
    threshold is lowered (samsim_set_launch_split) so that a 1 000-column ensemble splits.  Several steps back to back (no wait in between), output
    snapshot and status included, must equal the unsplit run bit for bit."""
    import samsim_amd
    from tests.shard_worker import ensemble_shard
    results = []
    for split in (0, 2):
        st, status = ensemble_shard(0, 1000, 260, launches=4, split_blocks=split)
        results.append((st, status))
    (a, xa), (b, xb) = results
    assert np.array_equal(xa, xb) and not xa.any()
    assert np.array_equal(a.lay[:10], b.lay[:10]) and np.array_equal(a.scal, b.scal) and np.array_equal(a.n_active, b.n_active)
