"""SURVEY.md section 8 f.1 on the GPU: device-side ensemble statistics, binary checkpoint / restart through the C-ABI
(Python mirror and Fortran host write and read the same stream format)."""
import os
import subprocess

import numpy as np
import pytest

import samsim_amd
from samsim_amd import checkpoint, testcases as tcs
from tests.helpers import golden, load_checkpoint, sheba_forcing, ROOT
from tests.oracle_lib import oracle_solver

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "host", "samsim_host.x")


def test_ensemble_stats_device_reduction_matches_numpy_and_checker():
    """70 001 columns (not a multiple of anything), perturbed SHEBA ensemble just after an output point; three corrupted
    columns stop with a reference STOP code and must be left out of the statistics"""
    ncol = 70001
    st1, clock = load_checkpoint("tc4_spunup_state.npz")
    cfg, _ = tcs.testcase4(1)
    st = st1.replicate(ncol)
    for c in (5, 40000, 70000):
        st.arr("H_abs")[0, c] = -1.0e15            # getT cannot converge -> STOP 99
    g = samsim_amd.hip_solver(cfg, ncol)
    g.set_forcing(*sheba_forcing(), *tcs.ensemble_perturbation(ncol))
    g.set_state(st)
    g.set_clock(**clock)
    g.run_to_output()
    status = g.get_status()[0]
    ok = status == 0
    assert (~ok).sum() == 3
    s = g.get_state()
    names = ["thickness", "thick_snow", "bulk_salin", "freeboard", "T_top", "T2m", "m_snow"]
    q = g.ensemble_stats(names + ["N_active"])
    for n in names:
        v = s.sc(n)[ok]
        assert q[n].count == ncol - 3
        assert q[n].min == v.min() and q[n].max == v.max(), n
        assert abs(q[n].mean - v.mean()) <= 1e-12 * max(1.0, abs(v.mean())), n
        assert abs(q[n].std - v.std()) <= 1e-10 * max(1e-3, v.std()), n
    assert q["N_active"].mean == pytest.approx(s.n_active[ok].mean(), rel=1e-14)
    assert q["thickness"].mean > 1.0 and q["T2m"].std > 1.0 and q["thick_snow"].std > 0.0
    # the checker library implements the same semantics: same numbers on a small slice of the same ensemble
    n2 = 64
    o = oracle_solver(cfg, n2)
    g2 = samsim_amd.hip_solver(cfg, n2)
    for x in (o, g2):
        x.set_forcing(*sheba_forcing(), *tcs.ensemble_perturbation(n2))
        x.set_state(st1.replicate(n2))
        x.set_clock(**clock)
        x.run_to_output()
    qo, qg = o.ensemble_stats(names), g2.ensemble_stats(names)
    for n in names:
        for f in ("mean", "min", "max", "std"):
            a, b = getattr(qg[n], f), getattr(qo[n], f)
            assert abs(a - b) <= 1e-6 * max(abs(b), 1e-3), (n, f, a, b)


def test_checkpoint_restart_continues_bitwise(tmp_path):
    """save -> new handle -> load: the restarted run is bit-identical to the uninterrupted one (full checkpoint), through
    an output point and with a column chunk that does not divide the ensemble"""
    ncol = 1000
    st1, clock = load_checkpoint("tc4_melt_state.npz")
    cfg, _ = tcs.testcase4(1)
    pert = tcs.ensemble_perturbation(ncol)

    def fresh():
        g = samsim_amd.hip_solver(cfg, ncol)
        g.set_forcing(*sheba_forcing(), *pert)
        return g
    a = fresh()
    a.set_state(st1.replicate(ncol))
    a.set_clock(**clock)
    a.step(3000)
    path = str(tmp_path / "ens.chk")
    checkpoint.save(a, path, chunk=384)
    assert os.path.getsize(path) > ncol * cfg.nlayer * 15 * 8
    b = fresh()
    h = checkpoint.load(b, path)
    assert h["step"] == a.get_clock().step
    a.step(7000)
    b.step(7000)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa.n_active, sb.n_active)
    assert np.array_equal(sa.lay[:4], sb.lay[:4]) and np.array_equal(sa.scal, sb.scal)
    assert len(np.unique(sa.arr("H_abs")[0])) > ncol // 2    # the perturbed members really differ
    # prognostic-only checkpoint: restart agrees to round-off (the diagnostics are rebuilt by the first sweep)
    checkpoint.save(a, path, narr=4)
    c = fresh()
    checkpoint.load(c, path)
    a.step(500)
    c.step(500)
    sa, sc = a.get_state(), c.get_state()
    assert np.array_equal(sa.n_active, sc.n_active)
    k = np.arange(cfg.nlayer)[:, None] < sa.n_active[None, :]
    for n in ("H_abs", "S_abs", "m", "thick"):
        x, y = sa.arr(n)[k], sc.arr(n)[k]
        assert np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-3)) <= 1e-9, n


def test_checkpoint_with_tracers_continues_bitwise(tmp_path):
    """testcase 6 as init ships it (tank_flag 2 + one tracer: the concentration of the water below is per-column state):
    save -> new handle -> set_tracers -> load is bit-identical to the uninterrupted run, tracers included"""
    ncol = 200
    cfg, st = tcs.testcase6(ncol)
    bottom, total, q = tcs.tracers(cfg, st)

    def fresh():
        g = samsim_amd.hip_solver(cfg, ncol)
        g.set_tracers(bottom, total)
        return g
    a = fresh()
    a.set_state(st)
    a.set_tracer_state(q)
    a.set_clock()
    a.step(20000)
    path = str(tmp_path / "bgc.chk")
    checkpoint.save(a, path, chunk=128)
    b = fresh()
    assert checkpoint.load(b, path)["n_bgc"] == 1
    a.step(5000)
    b.step(5000)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa.n_active, sb.n_active) and sa.n_active.min() > 3
    assert np.array_equal(sa.lay[:4], sb.lay[:4]) and np.array_equal(sa.scal, sb.scal)
    (qa, ba), (qb, bb) = a.get_tracer_state(), b.get_tracer_state()
    assert np.array_equal(qa, qb) and np.array_equal(ba, bb)
    assert ba.min() > bottom[0]                               # the tank's water got richer in tracer
    cfg0, _ = tcs.testcase6(ncol)
    with pytest.raises(ValueError):
        checkpoint.load(samsim_amd.hip_solver(cfg0, ncol), path)


@pytest.mark.skipif(not os.path.exists(HOST), reason="Fortran host not built (no flang)")
def test_fortran_host_restart_and_ensemble_file(tmp_path):
    """the Fortran host: run, write a restart file, continue from it in a second process; the rows printed after the restart
    equal those of an uninterrupted run; dat_ensemble.dat carries one statistics row per output point; the Python mirror
    reads the Fortran-written file"""
    def run(d, nml):
        d.mkdir()
        (d / "output").mkdir()
        (d / "samsim.nml").write_text(nml)
        r = subprocess.run([HOST], cwd=d, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout
    # init(1) as it ships: two passive tracers, which the restart file carries
    run(tmp_path / "full", "&samsim_run testcase=1, ncol=96, max_steps=12000 /\n")
    run(tmp_path / "part1", "&samsim_run testcase=1, ncol=96, max_steps=7300, restart_out='../tc1.chk' /\n")
    out = run(tmp_path / "part2", "&samsim_run testcase=1, ncol=96, max_steps=12000, restart_in='../tc1.chk' /\n")
    assert "restarted from" in out
    hdr = checkpoint.read_header(str(tmp_path / "tc1.chk"))
    assert (hdr["ncol"], hdr["nlayer"], hdr["step"], hdr["narr"], hdr["testcase"], hdr["n_bgc"]) == (96, 90, 7300, 15, 1, 2)
    full = (tmp_path / "full" / "output" / "dat_T.dat").read_text().splitlines()
    p1 = (tmp_path / "part1" / "output" / "dat_T.dat").read_text().splitlines()
    p2 = (tmp_path / "part2" / "output" / "dat_T.dat").read_text().splitlines()
    assert len(full) == 4 and len(p1) == 3 and len(p2) == 1        # outputs at steps 1, 3602, 7203, 10804
    assert p1 == full[:3] and p2 == full[3:]
    for name in ("dat_bgc01.bu.dat", "dat_bgc02.br.dat"):
        f, q1, q2 = ((tmp_path / d / "output" / name).read_text().splitlines() for d in ("full", "part1", "part2"))
        assert len(f) == 4 and q1 == f[:3] and q2 == f[3:], name
    ens = np.loadtxt(tmp_path / "full" / "output" / "dat_ensemble.dat")
    assert ens.shape == (4, 26) and (ens[:, 1] == 96).all()
    thick = np.loadtxt(tmp_path / "full" / "output" / "dat_vital_signs.dat")
    # identical columns: min == max; mean equal to them and std zero up to the rounding of a sum of 96 equal numbers (ES16.8)
    assert np.abs(ens[:, 3] - ens[:, 4]).max() == 0.0
    assert np.abs(ens[:, 2] - ens[:, 3]).max() == 0.0 and np.abs(ens[:, 5]).max() <= 1e-15
    assert thick.shape[0] == 4
