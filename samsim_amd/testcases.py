"""Host-side mirror of the reference's per-testcase setup `init(testcase)` (mo_init.f90:73-2034) for the
testcases in scope (1-7 and 9: every shipped testcase that runs without bgc tracers and lab snow) plus the synthetic BASELINE configurations.

`init` in the reference fills the `mo_data` globals; here it returns the POD `Config` that crosses the C-ABI
and the SoA initial `State`.  Defaults follow mo_init.f90:83-132, the common tail mo_init.f90:1981-2031.
"""
from __future__ import annotations

import os

import numpy as np

from .capi import Config, State, S, A, NSCAL, NPROG, NARR

RHO_L = 1028.0      # mo_parameters.f90:53
C_L = 3400.0        # mo_parameters.f90:51
RHO_SNOW = 330.0    # mo_parameters.f90:87
LATENT_HEAT = 333500.0
SIGMA = 5.6704 * float(np.float32(1e-8))   # mo_parameters.f90:59 (`1e-8` is a default-REAL literal)
LENGTH_INPUT = 13148  # mo_grotz.f90:132


def default_config() -> Config:
    """default flags, mo_init.f90:83-109, and mo_parameters.f90:107,110"""
    c = Config()
    c.struct_size = __import__("ctypes").sizeof(Config)
    c.boundflux_flag, c.atmoflux_flag, c.albedo_flag = 1, 1, 2
    c.grav_heat_flag, c.flush_heat_flag, c.flood_flag, c.flush_flag, c.grav_flag, c.harmonic_flag = 1, 1, 2, 5, 2, 2
    c.prescribe_flag, c.salt_flag = 1, 1
    c.turb_flag, c.bottom_flag, c.tank_flag = 2, 1, 1
    c.precip_flag, c.freeboard_snow_flag, c.snow_flush_flag, c.snow_precip_flag = 0, 0, 1, 1
    c.debug_flag, c.bgc_flag, c.lab_snow_flag = 1, 1, 0
    c.k_snow_flush, c.max_flux_plate = 0.75, 10000.0
    return c


def _finish(c: Config) -> Config:
    """common tail, mo_init.f90:1994-2001"""
    c.n_middle = c.nlayer - c.n_top - c.n_bottom
    c.thick_min = c.thick_0 / 2.0
    c.i_time_out = int(c.time_out / c.dt)
    return c


def i_time(c: Config) -> int:
    return int(c.time_total / c.dt)


def _blank_state(c: Config, ncol: int) -> State:
    """sub_allocate zeros (mo_init.f90:2081-2087) + defaults mo_init.f90:1982-1990"""
    st = State.empty(ncol, c.nlayer, NARR)
    st.arr("T")[:] = c.T_bottom
    st.arr("S_bu")[:] = c.S_bu_bottom
    st.arr("psi_l")[:] = 1.0
    st.sc("precip_scale")[:] = 1.0
    st.sc("S_bu_bottom")[:] = c.S_bu_bottom
    return st


def testcase1(ncol: int = 1):
    """mo_init.f90:865-945: cooling-plate tank experiment, T_top toggles -5/-10 every 12 h (bgc off)."""
    c = default_config()
    c.testcase = 1
    c.nlayer, c.n_top, c.n_bottom = 90, 5, 5
    c.turb_flag, c.boundflux_flag, c.grav_heat_flag, c.flush_flag, c.salt_flag = 1, 1, 1, 1, 2
    c.T_bottom, c.S_bu_bottom = -1.0, 34.0
    c.thick_0, c.dt, c.time_out = 0.002, 1.0, 3600.0
    c.time_total = c.time_out * 72.0
    _finish(c)
    st = _blank_state(c, ncol)
    st.sc("T_top")[:] = -5.0
    st.arr("thick")[0] = c.thick_0
    st.arr("m")[0] = st.arr("thick")[0] * RHO_L
    st.arr("S_abs")[0] = c.S_bu_bottom * st.arr("m")[0]
    st.arr("H_abs")[0] = st.arr("m")[0] * c.T_bottom * C_L
    return c, st


def testcase4(ncol: int = 1, nlayer: int = 100, n_top: int = 20, n_bottom: int = 20):
    """mo_init.f90:1127-1207: SHEBA / ERA-interim forced multi-year run (`Nlayer=100 = 20+60+20`)."""
    c = default_config()
    c.testcase = 4
    c.nlayer, c.n_top, c.n_bottom = nlayer, n_top, n_bottom
    c.atmoflux_flag, c.precip_flag, c.boundflux_flag = 2, 1, 2
    c.snow_flush_flag, c.flush_heat_flag, c.snow_precip_flag = 1, 2, 1
    c.T_bottom, c.S_bu_bottom = -1.0, 34.0
    c.thick_0, c.time_out, c.dt = 0.01, 86400.0, 10.0
    c.time_total = c.time_out * 365.0 * 4.5
    _finish(c)
    st = _blank_state(c, ncol)
    st.arr("thick")[0] = c.thick_0
    st.arr("m")[:] = st.arr("thick") * RHO_L
    st.arr("S_abs")[:] = c.S_bu_bottom * st.arr("m")
    st.arr("H_abs")[:] = 0.0
    return c, st


def tracers(cfg: Config, st: State):
    """the tracer set-up init(testcase) makes when it switches bgc on (testcase 1: mo_init.f90:921-942, 2: :1006-1020,
    6: :1333-1345): sets cfg.bgc_flag = 2 and returns (bgc_bottom, bgc_total or None, bgc_abs[n_bgc][nlayer][ncol]);
    the single initial water layer holds bgc_bottom * m(1)"""
    tc = cfg.testcase
    if tc == 1:
        bottom, total = np.array([400.0, 500.0]), None
    elif tc == 2:
        bottom = np.array([385.0, 385.0])
        total = bottom * RHO_L * 1.0              # bgc_bottom*rho_l*tank_depth, in this order
    elif tc == 6:
        bottom = np.array([385.0])
        total = bottom * RHO_L * 0.159
    else:
        raise ValueError(f"init({tc}) of the reference runs without tracers")
    cfg.bgc_flag = 2
    q = np.zeros((len(bottom), cfg.nlayer, st.ncol))
    q[:, 0, :] = bottom[:, None] * st.arr("m")[0][None, :]
    return bottom, total, q


def _tank(testcase, ncol, nlayer, n_top, n_bottom, tank_depth, alpha_stable, fl_q_bottom, T2m, T_top, T_bottom, S_bu_bottom,
          thick_0, dt, time_out, n_out):
    """the tank experiments (testcases 2, 6, 9): air temperature T2m over a tank of finite depth, boundflux_flag 3,
    tank_flag 2 (the salt the ice rejects raises the salinity of the water below)"""
    c = default_config()
    c.testcase = testcase
    c.nlayer, c.n_top, c.n_bottom = nlayer, n_top, n_bottom
    c.tank_flag, c.boundflux_flag, c.grav_heat_flag = 2, 3, 1
    c.alpha_flux_instable, c.alpha_flux_stable = 22.0, alpha_stable
    c.T_bottom, c.S_bu_bottom = T_bottom, S_bu_bottom
    c.thick_0, c.dt, c.time_out = thick_0, dt, time_out
    c.time_total = n_out
    c.m_total = RHO_L * tank_depth
    c.S_total = RHO_L * S_bu_bottom * tank_depth
    _finish(c)
    st = _blank_state(c, ncol)
    st.sc("fl_q_bottom")[:] = fl_q_bottom
    st.sc("T2m")[:] = T2m
    st.sc("T_top")[:] = T_top
    st.sc("S_bu_bottom")[:] = S_bu_bottom
    st.arr("thick")[0] = c.thick_0
    st.arr("m")[:] = st.arr("thick") * RHO_L
    st.arr("S_abs")[:] = c.S_bu_bottom * st.arr("m")
    st.arr("H_abs")[:] = st.arr("m") * c.T_bottom          # (sic: no c_l, mo_init.f90:1002)
    return c, st


def testcase2(ncol: int = 1):
    """mo_init.f90:948-1003: 1 m tank, T2m -20 C, warming after 15 and 25 days (sub_test2); bgc off"""
    return _tank(2, ncol, 100, 3, 10, 1.0, 15.0, 10.0, -20.0, -18.0, 0.0, 31.2, 0.01, 30.0, 3600.0 * 6.0,
                 3600.0 * 6.0 * 4.0 * 30.0)


def testcase6(ncol: int = 1):
    """mo_init.f90:1278-1330: 15.9 cm tank, thick_0 2.5 mm, dt 0.5 s, T2m switching between -18 and -5 C (sub_test6)"""
    return _tank(6, ncol, 40, 3, 3, 0.159, 11.0, 35.0, -18.0, -18.0, 0.0, 31.2, 0.0025, 0.5, 1800.0 / 2.0,
                 1800.0 / 2.0 * 39.0 * 2.0 * 2.0)


def testcase9(ncol: int = 1):
    """mo_init.f90:1684-1740: 0.8 m tank, freeze at -15 C for three days, then melt at +1 C (sub_test9)"""
    return _tank(9, ncol, 100, 3, 10, 0.8, 15.0, 10.0, -15.0, -10.0, -0.07, 34.6, 0.005, 10.0, 3600.0 * 2.0,
                 3600.0 * 2.0 * 12.0 * 6.0)


def testcase33(ncol: int = 1):
    """mo_init.f90:1779-1873: cooling-chamber experiment on nearly fresh water (S 0.13), 0.94 m tank, constant T2m -15 C"""
    return _tank(33, ncol, 100, 3, 10, 0.94, 15.0, 10.0, -15.0, -10.0, 0.5, 0.13, 0.005, 10.0, 60.0 * 5.0,
                 60.0 * 5.0 * 12.0 * 6.0)


def testcase34(ncol: int = 1):
    """mo_init.f90:1876-1975: the same chamber on sea water (S 34.9), ten days with the air temperature of sub_test34
    (0 C for two hours, -15 C to day 5, -5 C to day 7, then +1 C)"""
    return _tank(34, ncol, 100, 3, 10, 0.94, 15.0, 10.0, -15.0, -10.0, 0.5, 34.9, 0.005, 10.0, 60.0 * 10.0, 86400.0 * 10.0)


def testcase50(ncol: int = 1):
    """mo_init.f90:1497-1531: three years of growth under the Notz climatological fluxes from 5 mm of sea water, 70 layers;
    the spin-up of the convection study of Griewank & Notz 2012 (the reference's defaults otherwise: atmoflux_flag 1,
    boundflux_flag 2, gravity drainage, flushing, flooding)"""
    c = default_config()
    c.testcase = 50
    c.nlayer, c.n_top, c.n_bottom = 70, 5, 5
    c.atmoflux_flag, c.precip_flag, c.boundflux_flag = 1, 0, 2
    c.T_bottom, c.S_bu_bottom = -1.72, 34.0
    c.thick_0, c.dt, c.time_out = 0.005, 10.0, 3600.0 * 24.0 * 30.0
    c.time_total = c.time_out * 12.0 * 3.0
    _finish(c)
    st = _blank_state(c, ncol)
    st.sc("fl_q_bottom")[:] = 20.0
    st.sc("T_top")[:] = -20.0
    st.arr("thick")[0] = c.thick_0
    st.arr("m")[0] = st.arr("thick")[0] * RHO_L
    st.arr("S_abs")[0] = c.S_bu_bottom * st.arr("m")[0]
    st.arr("H_abs")[0] = st.arr("m")[0] * c.T_bottom * C_L
    return c, st


def testcase3(ncol: int = 1):
    """mo_init.f90:1045-1080: Notz climatological fluxes (atmoflux 1) + constant snow fall (sub_test3), 20 layers of 3 cm."""
    c = default_config()
    c.testcase = 3
    c.nlayer, c.n_top, c.n_bottom = 20, 5, 5
    c.atmoflux_flag, c.precip_flag, c.boundflux_flag = 1, 0, 2
    c.T_bottom, c.S_bu_bottom = -1.0, 34.0
    c.thick_0, c.dt, c.time_out = 0.03, 60.0, 86400.0 * 3.5
    c.time_total = c.time_out * 54.0 * 2.0 * 2.0
    _finish(c)
    st = _blank_state(c, ncol)
    st.sc("fl_q_bottom")[:] = 8.0
    st.arr("thick")[0] = c.thick_0
    st.arr("m")[:] = st.arr("thick") * RHO_L
    st.arr("S_abs")[:] = c.S_bu_bottom * st.arr("m")
    st.arr("H_abs")[:] = 0.0
    return c, st


def testcase5(ncol: int = 1):
    """mo_init.f90:1210-1273: 1 m slab of fresh-ish ice (all 100 layers active) warmed by a fixed flux (atmoflux 3),
    flushing only; the salinity is reset to 5 g/kg at step 2 (mo_grotz.f90:543-544)."""
    c = default_config()
    c.testcase = 5
    c.nlayer, c.n_top, c.n_bottom = 100, 20, 10
    c.boundflux_flag, c.atmoflux_flag, c.flush_heat_flag, c.flush_flag, c.grav_flag, c.flood_flag = 2, 3, 2, 5, 1, 1
    c.T_bottom, c.S_bu_bottom = 0.0, 5.0
    c.thick_0, c.dt, c.time_out = 0.01, 10.0, 3600.0 * 3.0
    c.time_total = c.time_out * 24.0 * 10.0
    _finish(c)
    st = _blank_state(c, ncol)
    st.n_active[:] = c.nlayer
    st.sc("fl_sw")[:] = 0.0
    st.sc("fl_rest")[:] = 290.0 ** 4 * SIGMA
    st.sc("fl_q_bottom")[:] = 15.0
    st.arr("thick")[:] = c.thick_0
    st.arr("m")[:] = st.arr("thick") * RHO_L
    st.arr("S_abs")[:] = st.arr("m") * c.S_bu_bottom
    st.arr("H_abs")[:] = st.arr("m") * (-90.0) * C_L
    return c, st


def testcase7(ncol: int = 1):
    """mo_init.f90:1360-1395: SHEBA forcing as testcase 4 with the simple parametrisations (albedo 1, grav 3, flush 4,
    flood 3), 9 years."""
    c = default_config()
    c.testcase = 7
    c.nlayer, c.n_top, c.n_bottom = 100, 20, 20
    c.atmoflux_flag, c.precip_flag, c.boundflux_flag = 2, 1, 2
    c.albedo_flag, c.grav_heat_flag, c.flush_heat_flag, c.flush_flag, c.grav_flag, c.flood_flag = 1, 2, 2, 4, 3, 3
    c.T_bottom, c.S_bu_bottom = -1.0, 34.0
    c.thick_0, c.time_out, c.dt = 0.01, 86400.0 / 2.0, 10.0
    c.time_total = c.time_out * 365.0 * 9.0
    _finish(c)
    st = _blank_state(c, ncol)
    st.arr("thick")[0] = c.thick_0
    st.arr("m")[:] = st.arr("thick") * RHO_L
    st.arr("S_abs")[:] = c.S_bu_bottom * st.arr("m")
    st.arr("H_abs")[:] = 0.0
    return c, st


def config5(ncol: int = 1, nlayer: int = 500):
    """SURVEY.md section 8(d) cfg5 (LDS-pressure / high-resolution configuration): testcase-4 flags (gravity drainage +
    flushing + flooding all active), Nlayer = 20 + (nlayer-40) + 20, thick_0 = 0.004 m, dt = 2 s
    (stability number k_s*dt/(rho_s*c_s*thick_0**2) = 0.15 < 0.5, cf. mo_grotz.f90:376), every layer active from the start:
    a brine-saturated slab with a linear -8 .. -1.9 C profile and S_bu = 5 g/kg (in the manner of testcase 5,
    mo_init.f90:1270-1273) under 0.7 m of snow, with the SHEBA tables started in the melt season (day 340).
    Returns (cfg, state, clock)."""
    c = default_config()
    c.testcase = 4
    c.nlayer, c.n_top, c.n_bottom = nlayer, 20, 20
    c.atmoflux_flag, c.precip_flag, c.boundflux_flag = 2, 1, 2
    c.snow_flush_flag, c.flush_heat_flag, c.snow_precip_flag = 1, 2, 1
    c.T_bottom, c.S_bu_bottom = -1.0, 34.0
    c.thick_0, c.time_out, c.dt = 0.004, 86400.0, 2.0
    c.time_total = c.time_out * 30.0
    _finish(c)
    st = _blank_state(c, ncol)
    st.n_active[:] = nlayer
    # linear temperature profile -8 C (top) .. -1.9 C (bottom), S_bu = 5 g/kg, brine-saturated (no gas, no excess volume):
    # H from the enthalpy relation getT inverts (mo_thermo_functions.f90:95), m from V_s + V_l = thick
    k = (np.arange(nlayer) + 0.5) / nlayer
    T = -8.0 + (8.0 - 1.9) * k
    S_br = -18.7 * T - 0.519 * T ** 2 - 0.00535 * T ** 3      # sea-salt liquidus, mo_thermo_functions.f90:324-326
    S_bu = 5.0
    phi = 1.0 - S_bu / S_br
    H = -LATENT_HEAT + LATENT_HEAT * S_bu / S_br + 2020.0 * T + 7.6973 * T * T / 2.0
    m = c.thick_0 / (phi / 920.0 + (1.0 - phi) / RHO_L)
    st.arr("thick")[:] = c.thick_0
    st.arr("m")[:] = m[:, None]
    st.arr("S_abs")[:] = (S_bu * m)[:, None]
    st.arr("H_abs")[:] = (H * m)[:, None]
    # 0.7 m of snow: heavier than the slab's buoyancy, so flooding fires as well as drainage and (melt season) flushing
    st.sc("thick_snow")[:] = 0.7
    st.sc("m_snow")[:] = 0.7 * RHO_SNOW
    st.sc("H_abs_snow")[:] = -st.sc("m_snow") * LATENT_HEAT
    st.sc("psi_s_snow")[:] = RHO_SNOW / 920.0
    day = 340
    clock = dict(time=day * 86400.0, step=int(day * 86400 / c.dt), n_time_out=0, time_counter=day * 8 + 1, n_outputs=day)
    return c, st, clock


def read_forcing(directory: str, length: int = LENGTH_INPUT):
    """sub_input (mo_functions.f90:304-327): list-directed read of the first `length` values of
    flux_sw/flux_lw/T2m/precip.txt.input; returns (fl_sw, fl_lw, T2m, precip)."""
    out = []
    for name in ("flux_sw", "flux_lw", "T2m", "precip"):
        a = np.loadtxt(os.path.join(directory, name + ".txt.input")).ravel()[:length]
        out.append(np.ascontiguousarray(a, dtype=np.float64))
    return tuple(out)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x.copy()
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def ensemble_perturbation(ncol: int, col0: int = 0):
    """dT2m_i in U(-2,2) K, precip scale 1+eps_i, eps_i in U(-0.3,0.3), from
    splitmix64(column_id xor 0x5A5A2026); global column 0 is unperturbed.  col0 = first global column id
    (multi-GPU shards regenerate their own slice)."""
    with np.errstate(over="ignore"):
        ids = np.arange(col0, col0 + ncol, dtype=np.uint64) ^ np.uint64(0x5A5A2026)
        h1 = splitmix64(ids)
        h2 = splitmix64(h1)
    u1 = (h1 >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    u2 = (h2 >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    dT = -2.0 + 4.0 * u1
    ps = 1.0 + (-0.3 + 0.6 * u2)
    if col0 == 0 and ncol > 0:
        dT[0] = 0.0
        ps[0] = 1.0
    return dT, ps
