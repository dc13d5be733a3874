"""Host-side mirror of mo_init (samsim_amd/testcases.py) and ensemble helpers; CPU only."""
import re

import numpy as np
import pytest

from samsim_amd import testcases as tcs
from samsim_amd.capi import State, NARR, NSCAL
from tests.helpers import golden
from tests.oracle_lib import oracle_solver


def test_testcase1_settings_match_reference_dat_settings():
    """dat_settings.dat of Reference_testcase1_with_Version_2 (the reference's echo of init, mo_output.f90:41-106)"""
    text = str(golden("tc1_reference_dat.npz")["settings_text"])
    kv = dict(re.findall(r"^(\w+)\s*=?\s+(-?[\d.]+)\s*$", text, flags=re.M))
    cfg, st = tcs.testcase1(1)
    assert float(kv["dt"]) == cfg.dt and float(kv["thick_0"]) == cfg.thick_0
    assert float(kv["time_out"]) == cfg.time_out and float(kv["time_total"]) == cfg.time_total
    assert float(kv["T_bottom"]) == cfg.T_bottom and float(kv["S_bu_bottom"]) == cfg.S_bu_bottom
    assert int(kv["N_top"]) == cfg.n_top and int(kv["N_middle"]) == cfg.n_middle and int(kv["N_bottom"]) == cfg.n_bottom
    assert int(kv["Nlayer"]) == cfg.nlayer
    for flag in ["boundflux_flag", "atmoflux_flag", "albedo_flag", "grav_flag", "flush_flag", "flood_flag",
                 "grav_heat_flag", "flush_heat_flag", "harmonic_flag", "prescribe_flag", "salt_flag", "turb_flag",
                 "bottom_flag", "tank_flag"]:
        assert int(kv[flag]) == getattr(cfg, flag), flag
    assert cfg.precip_flag == 0 and re.search(r"^precip_flag\s*$", text, flags=re.M)  # I9.0 prints a zero as blanks
    assert cfg.i_time_out == 3600 and tcs.i_time(cfg) == 259200 and cfg.thick_min == 0.001


def test_testcase4_settings():
    cfg, st = tcs.testcase4(3)
    assert (cfg.nlayer, cfg.n_top, cfg.n_middle, cfg.n_bottom) == (100, 20, 60, 20)
    assert (cfg.boundflux_flag, cfg.atmoflux_flag, cfg.precip_flag, cfg.flush_flag, cfg.flush_heat_flag) == (2, 2, 1, 5, 2)
    assert cfg.i_time_out == 8640 and tcs.i_time(cfg) == 14191200
    assert st.lay.shape == (NARR, 100, 3) and st.scal.shape == (NSCAL, 3)
    assert (st.arr("thick")[0] == 0.01).all() and (st.arr("thick")[1:] == 0).all()
    assert (st.arr("S_abs")[0] == 34.0 * (0.01 * 1028.0)).all() and (st.arr("H_abs") == 0).all()


def test_ensemble_perturbation_is_counter_based():
    dT, ps = tcs.ensemble_perturbation(1000)
    assert dT[0] == 0.0 and ps[0] == 1.0
    assert (np.abs(dT) <= 2.0).all() and (np.abs(ps - 1.0) <= 0.3).all()
    assert np.std(dT) > 1.0 and np.std(ps) > 0.15
    # a shard regenerates exactly its slice of the global sequence
    dT2, ps2 = tcs.ensemble_perturbation(100, col0=500)
    assert np.array_equal(dT2, dT[500:600]) and np.array_equal(ps2, ps[500:600])


def test_state_replicate_and_window():
    cfg, st = tcs.testcase1(1)
    r = st.replicate(5)
    assert r.lay.shape[2] == 5 and (r.lay == st.lay[:, :, :1]).all()
    w = r.window(2, 2)
    assert w.ncol == 2 and w.lay.flags.c_contiguous


def test_checkpoint_file_roundtrip_and_bitwise_continuation(tmp_path):
    """samsim_amd.checkpoint driven through the checker library: a run interrupted by save -> new handle
    -> load continues bit for bit, in column chunks that do not divide the ensemble"""
    from samsim_amd import checkpoint
    from tests.helpers import sheba_forcing
    ncol = 10
    cfg, st = tcs.testcase4(ncol)
    dT, ps = tcs.ensemble_perturbation(ncol)

    def fresh():
        o = oracle_solver(cfg, ncol)
        o.set_forcing(*sheba_forcing(), dT, ps)
        return o
    a = fresh()
    a.set_state(st)
    a.set_clock()
    a.step(9000)                       # past the first output point and the first ice layers
    path = str(tmp_path / "restart.chk")
    checkpoint.save(a, path, chunk=4)
    h = checkpoint.read_header(path)
    assert (h["ncol"], h["nlayer"], h["step"], h["testcase"]) == (ncol, cfg.nlayer, 9000, 4)
    b = fresh()
    checkpoint.load(b, path)
    a.step(3000)
    b.step(3000)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa.lay, sb.lay) and np.array_equal(sa.scal, sb.scal) and np.array_equal(sa.n_active, sb.n_active)
    ka, kb = a.get_clock(), b.get_clock()
    assert (ka.time, ka.step, ka.n_time_out, ka.time_counter) == (kb.time, kb.step, kb.n_time_out, kb.time_counter)
    with pytest.raises(ValueError):
        checkpoint.load(oracle_solver(cfg, ncol + 1), path)


def test_checkpoint_with_tracers_continues_bitwise(tmp_path):
    """a tank experiment with its tracer (init(6): bgc_flag 2, tank_flag 2, so that the concentration of the water below is
    per-column state): save -> new handle -> set_tracers -> load continues bit for bit; a handle without tracers refuses"""
    from samsim_amd import checkpoint
    ncol = 3
    cfg, st = tcs.testcase6(ncol)
    bottom, total, q = tcs.tracers(cfg, st)

    def fresh():
        o = oracle_solver(cfg, ncol)
        o.set_tracers(bottom, total)
        return o
    a = fresh()
    a.set_state(st)
    a.set_tracer_state(q)
    a.set_clock()
    a.step(20000)
    assert a.get_state().n_active.min() > 3
    path = str(tmp_path / "restart_bgc.chk")
    checkpoint.save(a, path, chunk=2)
    assert checkpoint.read_header(path)["n_bgc"] == 1
    b = fresh()
    checkpoint.load(b, path)
    a.step(5000)
    b.step(5000)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa.lay, sb.lay) and np.array_equal(sa.scal, sb.scal) and np.array_equal(sa.n_active, sb.n_active)
    (qa, ba), (qb, bb) = a.get_tracer_state(), b.get_tracer_state()
    assert np.array_equal(qa, qb) and np.array_equal(ba, bb)
    assert ba.min() > bottom[0] and qa[0, 2].min() > 0.0        # the tank got richer, the ice holds tracer
    cfg0, _ = tcs.testcase6(ncol)
    with pytest.raises(ValueError):
        checkpoint.load(oracle_solver(cfg0, ncol), path)


def test_ensemble_statistics_of_the_checker():
    """samsim_get_ensemble_stats semantics (count / mean / min / max / population std over the columns without a STOP
    code) on the checker library against numpy"""
    from tests.helpers import sheba_forcing
    ncol = 24
    cfg, st = tcs.testcase4(ncol)
    o = oracle_solver(cfg, ncol)
    o.set_forcing(*sheba_forcing(), *tcs.ensemble_perturbation(ncol))
    o.set_state(st)
    o.set_clock()
    o.run_to_output()
    o.step(8641)                       # second output point: vital signs of an ice-covered ensemble
    s = o.get_state()
    stats = o.ensemble_stats(["thickness", "thick_snow", "T2m", "N_active"])
    for n in ("thickness", "thick_snow", "T2m"):
        v = s.sc(n)
        q = stats[n]
        assert q.count == ncol and q.min == v.min() and q.max == v.max()
        assert abs(q.mean - v.mean()) <= 1e-13 * max(1.0, abs(v.mean())) and abs(q.std - v.std()) <= 1e-12 * max(1.0, v.std())
    assert stats["N_active"].max == s.n_active.max() and stats["N_active"].mean == pytest.approx(s.n_active.mean())
    assert stats["T2m"].std > 0.5


def test_device_layout_index_is_a_bijection(tmp_path):
    """samsim_device.h: the layer block is stored per 64-column block, [block][layer][array][lane].  DEV_LAY_INDEX must map every
    (array, layer, column) of a handle to its own double inside DEV_LAY_DOUBLES, keep a wave's 64 columns of one (array, layer)
    contiguous, and keep the sixteen arrays of a layer row within 8 KiB of each other (the kernel addresses them with the
    signed 13-bit immediate of one row address) -- checked on the host with the header compiled by g++."""
    import os
    import subprocess
    from tests.helpers import ROOT
    src = tmp_path / "layout.cpp"
    src.write_text(r'''
#include <cstdio>
#include <vector>
#include "samsim_device.h"
int main() {
  const int N = 7; const size_t ncol = 200;   // not a multiple of 64: the last block is ragged
  const size_t total = DEV_LAY_DOUBLES(N, ncol);
  std::vector<char> seen(total, 0);
  for (int a = 0; a < DEV_NARR; ++a) for (int k = 0; k < N; ++k) for (size_t c = 0; c < ncol; ++c) {
    const size_t i = DEV_LAY_INDEX(a, k, c, N, ncol);
    if (i >= total || seen[i]) { std::printf("clash a=%d k=%d c=%zu\n", a, k, c); return 1; }
    seen[i] = 1;
  }
  // lanes contiguous, arrays of a row 512 B apart, rows DEV_ROWB apart
  if (DEV_LAY_INDEX(3, 2, 65, N, ncol) != DEV_LAY_INDEX(3, 2, 64, N, ncol) + 1) return 2;
  if ((DEV_LAY_INDEX(4, 2, 64, N, ncol) - DEV_LAY_INDEX(3, 2, 64, N, ncol)) * 8 != 512) return 3;
  if ((DEV_LAY_INDEX(0, 3, 64, N, ncol) - DEV_LAY_INDEX(0, 2, 64, N, ncol)) * 8 != DEV_ROWB) return 4;
  if (DEV_ROWB != 8192 || DEV_NARR != 16) return 5;
  std::printf("ok %zu\n", total);
  return 0;
}
''')
    exe = tmp_path / "layout"
    inc = os.path.join(ROOT, "samsim_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout


def test_checkpoint_layout_without_the_status_block_is_refused(tmp_path):
    """the chunk layout changed when checkpoints began to carry the STOP codes: files of the older layout carry another magic
    ("SAMCHK01") and are refused, not mis-read"""
    import struct
    from samsim_amd import checkpoint
    p = tmp_path / "old.chk"
    p.write_bytes(struct.pack("<6q d 4q 5q", int.from_bytes(b"SAMCHK01", "little"), 4, 90, 15, 38, 1, 0.0, 0, 0, 1, 0, 0, 0, 0, 0, 0))
    with pytest.raises(ValueError, match="SAMCHK01"):
        checkpoint.read_header(str(p))
    p.write_bytes(struct.pack("<6q d 4q 5q", checkpoint.MAGIC, 4, 90, 15, 38, 1, 0.0, 7, 0, 1, 0, 0, 1, 0, 0, 0))
    assert checkpoint.read_header(str(p))["step"] == 7


def test_plain_form_of_the_permeability_power_against_the_exact_power(tmp_path):
    """samsim_pow.h compiled for the host: x**3.1 as x*x*x * exp(0.1 * log x) (the form the kernel keeps for liquid fractions below
    1e-33, and the reference of its tenth-root form on the GPU, tools/div_probe) against powl(x, 3 + 0.1) over the range of the
    permeability law, 1e-9 .. 3000: within 1e-15 relative"""
    import os
    import subprocess
    src = tmp_path / "powtest.c"
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "samsim_amd", "csrc", "samsim_pow.h")
    src.write_text('''#include <stdio.h>
#include <math.h>
#include "%s"
int main(void) {
  double mx = 0.0;
  const long double e = (long double)3.0 + (long double)0.1;   /* the exponent the form evaluates: 3 + the double 0.1 */
  for (int i = 0; i < 4000000; i++) {
    const double u = (i + 0.5) / 4000000.0, x = exp(log(1e-9) + u * (log(3000.0) - log(1e-9)));
    const double r = (double)powl((long double)x, e), err = fabs(sp_pow_3p1_plain(x) - r) / r;
    if (err > mx) mx = err;
  }
  printf("%%.3e\\n", mx);
  return 0;
}
''' % hdr)
    exe = tmp_path / "powtest"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"])
    worst = float(subprocess.check_output([str(exe)]).decode())
    assert worst <= 1e-15, worst


def test_short_forms_of_the_scalar_powers_against_the_exact_powers(tmp_path):
    """samsim_pow.h compiled for the host: x*sqrt(x) for x**1.5 (bulk salinities 1e-3 .. 300 g/kg) and (x*x)*(x*x) for x**4
    (surface temperatures 150 .. 400 K) against powl: within 2 ulp (observed 2.2e-16 and 4.4e-16 relative)"""
    import os
    import subprocess
    src = tmp_path / "powtest2.c"
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "samsim_amd", "csrc", "samsim_pow.h")
    src.write_text('''#include <stdio.h>
#include <math.h>
#include "%s"
int main(void) {
  double m15 = 0.0, m4 = 0.0;
  for (int i = 0; i < 4000000; i++) {
    const double u = (i + 0.5) / 4000000.0, x = exp(log(1e-3) + u * (log(300.0) - log(1e-3))), t = 150.0 + 250.0 * u;
    const double r = (double)powl((long double)x, 1.5L), e = fabs(sp_pow_1p5(x) - r) / r;
    const double r4 = (double)powl((long double)t, 4.0L), e4 = fabs(sp_pow_4(t) - r4) / r4;
    if (e > m15) m15 = e;
    if (e4 > m4) m4 = e4;
  }
  printf("%%.3e %%.3e\\n", m15, m4);
  return 0;
}
''' % hdr)
    exe = tmp_path / "powtest2"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"])
    w15, w4 = (float(v) for v in subprocess.check_output([str(exe)]).decode().split())
    assert w15 <= 3.4e-16 and w4 <= 4.5e-16, (w15, w4)
