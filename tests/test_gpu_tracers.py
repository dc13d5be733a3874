"""Passive tracers (bgc_flag 2) on the GPU: the sparse restatement of fl_brine_bgc / bgc_advection in
the kernel against the reference's own records (fixtures) and against the oracle, which runs the reference's (N+1)^2
loops as written and is bit-identical to the flang build on testcases 1, 2 and 6 (tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest

import samsim_amd
from samsim_amd import testcases as tcs
from tests.helpers import RTOL, golden, load_checkpoint, rel_err, sheba_forcing
from tests.oracle_lib import oracle_solver

pytestmark = pytest.mark.gpu


def _setup(s, cfg, st, bottom, total, q, forcing=None, clock=None):
    if forcing is not None:
        s.set_forcing(*forcing)
    s.set_tracers(bottom, total)
    s.set_state(st)
    s.set_tracer_state(q)
    s.set_clock(**(clock or {}))


@pytest.mark.parametrize("tc,nout", [(1, 24), (2, 120), (6, 60)])   # (testcase 1 runs to its end in tests/test_gpu_fortran_host.py)
def test_tracers_of_the_shipped_testcases(tc, nout):
    """testcases 1, 2, 6 exactly as init() ships them (tracers on): bgc_abs and bgc_bottom of the HIP path at every output
    point against the reference's records; expulsion, gravity drainage with return flow, flushing (testcase 2 melts),
    bottom mixing, regridding, the tank's tracer budget"""
    ncol = 4
    cfg, st = getattr(tcs, f"testcase{tc}")(ncol)
    bottom, total, q = tcs.tracers(cfg, st)
    g = samsim_amd.hip_solver(cfg, ncol)
    _setup(g, cfg, st, bottom, total, q)
    g.set_output_window(1, 2)
    ref = golden(f"tc{tc}_bgc_ref.npz")
    for i in range(nout):
        out = g.run_to_output()
        a, b = g.get_tracer_output()
        assert out.step == ref["step"][i] and out.n_active[0] == ref["N_active"][i], f"tc{tc} output {i}"
        na = int(out.n_active[0])
        for w in range(2):
            assert rel_err(a[:, :na, w], ref["bgc_abs"][i][:, :na], 1e-6) <= RTOL, f"tc{tc} output {i}: bgc_abs"
            assert rel_err(b[:, w], ref["bgc_bottom"][i], 1e-6) <= RTOL, f"tc{tc} output {i}: bgc_bottom"
    assert not g.get_status()[0].any()
    a, b = g.get_tracer_state()
    assert (a == a[:, :, :1]).all()                   # identical columns stayed identical
    if total is not None:
        assert b.max() > bottom.max() + 1.0


def test_tracers_through_melt_and_flooding_against_the_oracle():
    """no shipped testcase combines tracers with flooding or with SHEBA's melt season, so these are checked against the
    oracle's literal restatement: three tracers with a depth-dependent initial profile, (i) the melt-onset checkpoint of
    the SHEBA run (flush3 with vertical and horizontal tracer fluxes, top melt regrids), (ii) cfg5 in small (120 layers would
    abort at once; 200 layers) with the snow load that drives flooding"""
    bottom = np.array([400.0, 500.0, 1.0])
    for which in ("melt", "flood"):
        if which == "melt":
            st, clock = load_checkpoint("tc4_melt_state.npz")
            cfg, _ = tcs.testcase4(1)
            steps = (1, 1000, 7641)
        else:
            cfg, st, clock = tcs.config5(1, nlayer=500)
            steps = (1, 300)
        cfg.bgc_flag = 2
        ncol = 3
        stn = st.replicate(ncol)
        q = bottom[:, None, None] * stn.arr("m")[None, :, :] * np.linspace(0.2, 1.0, cfg.nlayer)[None, :, None]
        g, o = samsim_amd.hip_solver(cfg, ncol), oracle_solver(cfg, ncol)
        for s in (g, o):
            _setup(s, cfg, stn, bottom, None, q, sheba_forcing(), clock)
        tot0 = q[:, :, 0].sum(axis=1)
        for n in steps:
            g.step(n)
            o.step(n)
            assert np.array_equal(g.get_status()[0], o.get_status()[0]) and not o.get_status()[0].any()
            (a, ab), (b, bb) = g.get_tracer_state(), o.get_tracer_state()
            so = o.get_state()
            na = int(so.n_active[0])
            assert np.array_equal(g.get_state().n_active, so.n_active)
            assert rel_err(a[:, :na], b[:, :na], 1e-9) <= RTOL, f"{which} +{n}: tracers"
        assert np.abs(b[:, :, 0].sum(axis=1) - tot0).max() > 1.0       # brine really moved tracer in or out of the column
