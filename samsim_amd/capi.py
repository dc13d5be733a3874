"""ctypes view of the C-ABI declared in include/samsim.h.

The struct layouts and call signatures are those of the product library
(``samsim_amd/csrc/libsamsim_hip.so``, symbol prefix ``samsim_``).  ``Solver`` is parameterised by the library
object and the symbol prefix so that the test-suite can drive a checker library that exports the same interface
under another prefix; this module itself only ever describes the interface (``samsim_amd.load()`` loads the HIP
library and fails loudly when it is missing -- there is no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

ABI_VERSION = 5

_CFG_INT_FIELDS = [
    "struct_size", "testcase", "nlayer", "n_top", "n_middle", "n_bottom",
    "atmoflux_flag", "grav_flag", "prescribe_flag", "grav_heat_flag", "flush_heat_flag", "turb_flag",
    "salt_flag", "boundflux_flag", "flush_flag", "flood_flag", "bottom_flag", "debug_flag", "precip_flag",
    "harmonic_flag", "tank_flag", "albedo_flag", "lab_snow_flag", "freeboard_snow_flag", "snow_flush_flag",
    "snow_precip_flag", "bgc_flag", "i_time_out",
]
_CFG_DBL_FIELDS = [
    "dt", "thick_0", "thick_min", "T_bottom", "S_bu_bottom", "k_snow_flush", "max_flux_plate", "time_out",
    "time_total", "alpha_flux_instable", "alpha_flux_stable", "m_total", "S_total",
]


class Config(C.Structure):
    """samsim_config (include/samsim.h); flags of mo_data.f90:136-155."""
    _fields_ = [(n, C.c_int32) for n in _CFG_INT_FIELDS] + [(n, C.c_double) for n in _CFG_DBL_FIELDS]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class StateSoA(C.Structure):
    _fields_ = [("ncol", C.c_int64), ("nlayer", C.c_int32), ("narr", C.c_int32),
                ("lay", C.POINTER(C.c_double)), ("scal", C.POINTER(C.c_double)),
                ("n_active", C.POINTER(C.c_int32))]


class Clock(C.Structure):
    _fields_ = [("time", C.c_double), ("step", C.c_int64), ("n_time_out", C.c_int32),
                ("time_counter", C.c_int32), ("n_outputs", C.c_int64)]


class Stat(C.Structure):
    """samsim_stat"""
    _fields_ = [("count", C.c_int64), ("mean", C.c_double), ("min", C.c_double), ("max", C.c_double), ("std", C.c_double)]


class OutputSoA(C.Structure):
    _fields_ = [("ncols", C.c_int64), ("nlayer", C.c_int32), ("reserved", C.c_int32),
                ("lay", C.POINTER(C.c_double)), ("scal", C.POINTER(C.c_double)),
                ("n_active", C.POINTER(C.c_int32)), ("time", C.c_double), ("step", C.c_int64)]


# enum samsim_scalar
SCALARS = [
    "m_snow", "H_abs_snow", "S_abs_snow", "thick_snow", "psi_s_snow", "psi_l_snow", "psi_g_snow", "T_snow",
    "phi_s", "T_top", "melt_thick", "T2m", "liquid_precip", "solid_precip", "fl_q_bottom",
    "grav_drain", "grav_salt", "grav_temp", "melt_out1", "melt_out2", "melt_out3", "melt_err",
    "freeboard", "T_freeze", "albedo", "fl_sw", "fl_lw", "melt_thick_snow", "fl_Q_snow",
    "energy_stored", "freshwater", "total_resist", "thickness", "bulk_salin", "fl_rest", "S_bu_bottom", "dT2m", "precip_scale",
]
S = {n: i for i, n in enumerate(SCALARS)}
NSCAL = len(SCALARS)
# enum samsim_layer_array
ARRAYS = ["H_abs", "S_abs", "m", "thick", "T", "phi", "psi_s", "psi_l", "psi_g", "S_bu", "S_br", "ray", "perm",
          "flush_v", "flush_h"]
A = {n: i for i, n in enumerate(ARRAYS)}
NARR = len(ARRAYS)
NPROG = 4

ERRORS = {0: "ok", -1: "bad argument", -2: "unsupported flag value", -3: "HIP error", -4: "no HIP device",
          -5: "no output snapshot", -6: "ABI mismatch", -7: "out of memory"}


class SamsimError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what}: error {code} ({ERRORS.get(code, '?')})")
        self.code = code


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


@dataclass
class State:
    """Host copy of the SoA column state: lay[narr, nlayer, ncol], scal[NSCAL, ncol], n_active[ncol]."""
    lay: np.ndarray
    scal: np.ndarray
    n_active: np.ndarray

    @property
    def ncol(self):
        return self.lay.shape[2]

    @property
    def nlayer(self):
        return self.lay.shape[1]

    def arr(self, name):
        return self.lay[A[name]]

    def sc(self, name):
        return self.scal[S[name]]

    def copy(self):
        return State(self.lay.copy(), self.scal.copy(), self.n_active.copy())

    def replicate(self, ncol):
        """state of column 0 repeated ncol times"""
        return State(np.ascontiguousarray(np.repeat(self.lay[:, :, :1], ncol, axis=2)),
                     np.ascontiguousarray(np.repeat(self.scal[:, :1], ncol, axis=1)),
                     np.ascontiguousarray(np.repeat(self.n_active[:1], ncol)))

    def window(self, c0, n):
        return State(np.ascontiguousarray(self.lay[:, :, c0:c0 + n]), np.ascontiguousarray(self.scal[:, c0:c0 + n]),
                     np.ascontiguousarray(self.n_active[c0:c0 + n]))

    def _c(self):
        assert self.lay.dtype == np.float64 and self.lay.flags.c_contiguous
        assert self.scal.dtype == np.float64 and self.scal.flags.c_contiguous and self.scal.shape == (NSCAL, self.ncol)
        assert self.n_active.dtype == np.int32 and self.n_active.flags.c_contiguous
        return StateSoA(self.ncol, self.nlayer, self.lay.shape[0], _dp(self.lay), _dp(self.scal), _ip(self.n_active))

    @staticmethod
    def empty(ncol, nlayer, narr=NARR):
        return State(np.zeros((narr, nlayer, ncol)), np.zeros((NSCAL, ncol)), np.ones(ncol, dtype=np.int32))


@dataclass
class Output:
    lay: np.ndarray
    scal: np.ndarray
    n_active: np.ndarray
    time: float
    step: int

    def arr(self, name):
        return self.lay[A[name]]

    def sc(self, name):
        return self.scal[S[name]]


class Solver:
    """One handle of the C-ABI of include/samsim.h."""

    def __init__(self, lib: C.CDLL, prefix: str, cfg: Config, ncol: int, device: int = 0):
        self._lib = lib
        self._p = prefix
        self.cfg = cfg
        self.ncol = int(ncol)
        self.nlayer = int(cfg.nlayer)
        self._h = C.c_void_p()
        self._bind()
        self._chk(self._create(device), "create")
        self._out_window = (0, 1)

    def _create(self, device):
        f = self._f("create")
        f.argtypes, f.restype = [C.POINTER(Config), C.c_int64, C.c_int32, C.POINTER(C.c_void_p)], C.c_int
        return f(C.byref(self.cfg), C.c_int64(self.ncol), C.c_int32(device), C.byref(self._h))

    # -- plumbing
    def _f(self, name):
        return getattr(self._lib, self._p + name)

    def _chk(self, rc, what):
        if rc != 0:
            raise SamsimError(rc, self._p + what)

    def _bind(self):
        vp, i64, i32, dp = C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double)
        sig = {
            "set_forcing": [vp, i32, dp, dp, dp, dp, dp, dp],
            "set_forcing_sites": [vp, i32, i32, dp, dp, dp, dp, C.POINTER(C.c_int32), dp, dp],
            "set_ocean": [vp, dp, dp],
            "set_state": [vp, C.POINTER(StateSoA), i64], "get_state": [vp, C.POINTER(StateSoA), i64],
            "set_clock": [vp, C.POINTER(Clock)], "get_clock": [vp, C.POINTER(Clock)],
            "step": [vp, i64], "set_output_window": [vp, i64, i64], "get_output": [vp, C.POINTER(OutputSoA)],
            "get_status": [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)],
            "set_status": [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), i64, i64],
            "get_work": [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
            "get_ensemble_stats": [vp, i32, C.POINTER(C.c_int32), C.POINTER(Stat)],
            "set_tracers": [vp, i32, dp, dp], "set_tracer_state": [vp, dp, i64, i64], "set_tracer_bottom": [vp, dp, i64, i64],
            "get_tracer_state": [vp, dp, dp, i64, i64], "get_tracer_output": [vp, dp, dp],
        }
        for n, a in sig.items():
            f = self._f(n)
            f.argtypes, f.restype = a, C.c_int
        f = self._f("steps_to_output")
        f.argtypes, f.restype = [vp], C.c_int64
        f = self._f("destroy")
        f.argtypes, f.restype = [vp], None
        self._bind_extra()

    def _bind_extra(self):
        vp, i64 = C.c_void_p, C.c_int64
        f = self._f("step_timed")
        f.argtypes, f.restype = [vp, i64, C.POINTER(C.c_double)], C.c_int
        f = self._f("steps_timed")
        f.argtypes, f.restype = [vp, i64, C.c_int32, C.POINTER(C.c_double)], C.c_int
        f = self._f("synchronize")
        f.argtypes, f.restype = [vp], C.c_int
        f = self._f("get_device")
        f.argtypes, f.restype = [vp, C.POINTER(C.c_int32), C.c_char_p, C.c_int32], C.c_int
        f = self._f("set_launch_split")
        f.argtypes, f.restype = [vp, i64, C.c_int32], C.c_int

    # -- API
    def set_forcing(self, fl_sw, fl_lw, T2m, precip, dT2m=None, precip_scale=None):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (fl_sw, fl_lw, T2m, precip)]
        n = len(arrs[0])
        assert all(len(a) == n for a in arrs)
        d = None if dT2m is None else np.ascontiguousarray(dT2m, dtype=np.float64)
        p = None if precip_scale is None else np.ascontiguousarray(precip_scale, dtype=np.float64)
        assert d is None or d.shape == (self.ncol,)
        assert p is None or p.shape == (self.ncol,)
        self._chk(self._f("set_forcing")(self._h, n, *[_dp(a) for a in arrs],
                                         _dp(d) if d is not None else None, _dp(p) if p is not None else None),
                  "set_forcing")

    def set_forcing_sites(self, fl_sw, fl_lw, T2m, precip, site_of_column, dT2m=None, precip_scale=None):
        """tables [nsites][len] (samsim_set_forcing_sites): column c reads set site_of_column[c]"""
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (fl_sw, fl_lw, T2m, precip)]
        nsites, n = arrs[0].shape
        assert all(a.shape == (nsites, n) for a in arrs)
        site = np.ascontiguousarray(site_of_column, dtype=np.int32)
        assert site.shape == (self.ncol,)
        d = None if dT2m is None else np.ascontiguousarray(dT2m, dtype=np.float64)
        p = None if precip_scale is None else np.ascontiguousarray(precip_scale, dtype=np.float64)
        self._chk(self._f("set_forcing_sites")(self._h, nsites, n, *[_dp(a) for a in arrs], _ip(site),
                                               _dp(d) if d is not None else None, _dp(p) if p is not None else None),
                  "set_forcing_sites")

    def set_ocean(self, dfl_q_bottom=None, S_bu_bottom=None):
        """the water below a grid of columns (samsim_set_ocean): per-column offset on the oceanic heat flux sub_test4 sets every
        step, per-column salinity of the water below (tank_flag 1); None switches a part off"""
        d = None if dfl_q_bottom is None else np.ascontiguousarray(dfl_q_bottom, dtype=np.float64)
        s = None if S_bu_bottom is None else np.ascontiguousarray(S_bu_bottom, dtype=np.float64)
        assert d is None or d.shape == (self.ncol,)
        assert s is None or s.shape == (self.ncol,)
        self._chk(self._f("set_ocean")(self._h, _dp(d) if d is not None else None, _dp(s) if s is not None else None), "set_ocean")

    def set_state(self, st: State, col0: int = 0):
        s = st._c()
        self._chk(self._f("set_state")(self._h, C.byref(s), col0), "set_state")

    def get_state(self, col0: int = 0, ncols: int | None = None, narr: int = NARR) -> State:
        n = self.ncol - col0 if ncols is None else ncols
        st = State.empty(n, self.nlayer, narr)
        s = st._c()
        self._chk(self._f("get_state")(self._h, C.byref(s), col0), "get_state")
        return st

    def set_clock(self, time=0.0, step=0, n_time_out=0, time_counter=1, n_outputs=0):
        c = Clock(time, step, n_time_out, time_counter, n_outputs)
        self._chk(self._f("set_clock")(self._h, C.byref(c)), "set_clock")

    def get_clock(self) -> Clock:
        c = Clock()
        self._chk(self._f("get_clock")(self._h, C.byref(c)), "get_clock")
        return c

    def step(self, nsteps: int = 1):
        self._chk(self._f("step")(self._h, nsteps), "step")

    def step_timed(self, nsteps: int) -> float:
        ms = C.c_double()
        self._chk(self._f("step_timed")(self._h, nsteps, C.byref(ms)), "step_timed")
        return ms.value

    def steps_timed(self, nsteps: int, nlaunches: int) -> float:
        """nlaunches launches of nsteps steps each, enqueued back to back; device time of the whole sequence [ms]"""
        ms = C.c_double()
        self._chk(self._f("steps_timed")(self._h, nsteps, nlaunches, C.byref(ms)), "steps_timed")
        return ms.value

    def synchronize(self):
        self._chk(self._f("synchronize")(self._h), "synchronize")

    def get_device(self):
        """(HIP device ordinal, PCI bus id) of the GPU this handle lives on"""
        dev, buf = C.c_int32(-1), C.create_string_buffer(32)
        self._chk(self._f("get_device")(self._h, C.byref(dev), buf, 32), "get_device")
        return int(dev.value), buf.value.decode()

    def set_launch_split(self, min_blocks: int = 8192, first_part_eighths: int = 4):
        """from how many 64-column blocks a step runs as two concurrent launches (0 = never), and the first part's share"""
        self._chk(self._f("set_launch_split")(self._h, min_blocks, first_part_eighths), "set_launch_split")

    def steps_to_output(self) -> int:
        return int(self._f("steps_to_output")(self._h))

    def set_output_window(self, col0: int, ncols: int):
        self._chk(self._f("set_output_window")(self._h, col0, ncols), "set_output_window")
        self._out_window = (col0, ncols)

    def get_output(self) -> Output:
        n = self._out_window[1]
        lay = np.zeros((NARR, self.nlayer, n))
        scal = np.zeros((NSCAL, n))
        na = np.zeros(n, dtype=np.int32)
        o = OutputSoA(n, self.nlayer, 0, _dp(lay), _dp(scal), _ip(na), 0.0, 0)
        self._chk(self._f("get_output")(self._h, C.byref(o)), "get_output")
        return Output(lay, scal, na, o.time, o.step)

    def get_status(self):
        st = np.zeros(self.ncol, dtype=np.int32)
        sp = np.zeros(self.ncol, dtype=np.int64)
        ly = np.zeros(self.ncol, dtype=np.int32)
        self._chk(self._f("get_status")(self._h, _ip(st), sp.ctypes.data_as(C.POINTER(C.c_int64)), _ip(ly)), "get_status")
        return st, sp, ly

    def set_status(self, status, step=None, layer=None, col0: int = 0):
        """restart: put back the STOP codes (and the step / layer of the failure) samsim_get_status returned"""
        st = np.ascontiguousarray(status, dtype=np.int32)
        sp = None if step is None else np.ascontiguousarray(step, dtype=np.int64)
        ly = None if layer is None else np.ascontiguousarray(layer, dtype=np.int32)
        self._chk(self._f("set_status")(self._h, _ip(st), sp.ctypes.data_as(C.POINTER(C.c_int64)) if sp is not None else None,
                                        _ip(ly) if ly is not None else None, col0, len(st)), "set_status")

    def get_work(self):
        a, b = C.c_int64(), C.c_int64()
        self._chk(self._f("get_work")(self._h, C.byref(a), C.byref(b)), "get_work")
        return a.value, b.value

    # -- passive tracers (bgc_flag 2)
    def set_tracers(self, bgc_bottom, bgc_total=None):
        """number of tracers = len(bgc_bottom); concentrations below the ice; tank totals for tank_flag 2"""
        b = np.ascontiguousarray(bgc_bottom, dtype=np.float64)
        t = None if bgc_total is None else np.ascontiguousarray(bgc_total, dtype=np.float64)
        self.n_bgc = len(b)
        self._chk(self._f("set_tracers")(self._h, self.n_bgc, _dp(b), _dp(t) if t is not None else None), "set_tracers")

    def set_tracer_state(self, bgc_abs, col0: int = 0):
        """bgc_abs[n_bgc][nlayer][ncols]"""
        a = np.ascontiguousarray(bgc_abs, dtype=np.float64)
        assert a.shape[:2] == (self.n_bgc, self.nlayer)
        self._chk(self._f("set_tracer_state")(self._h, _dp(a), col0, a.shape[2]), "set_tracer_state")

    def set_tracer_bottom(self, bgc_bottom, col0: int = 0):
        """bgc_bottom[n_bgc][ncols]: per-column concentration below the ice, as get_tracer_state returned it (restart)"""
        b = np.ascontiguousarray(bgc_bottom, dtype=np.float64)
        assert b.shape[0] == self.n_bgc
        self._chk(self._f("set_tracer_bottom")(self._h, _dp(b), col0, b.shape[1]), "set_tracer_bottom")

    def get_tracer_state(self, col0: int = 0, ncols: int | None = None):
        n = self.ncol - col0 if ncols is None else ncols
        a, b = np.zeros((self.n_bgc, self.nlayer, n)), np.zeros((self.n_bgc, n))
        self._chk(self._f("get_tracer_state")(self._h, _dp(a), _dp(b), col0, n), "get_tracer_state")
        return a, b

    def get_tracer_output(self):
        """tracer snapshot of the output window at the reference's output point: (bgc_abs, bgc_bottom)"""
        w = self._out_window[1]
        a, b = np.zeros((self.n_bgc, self.nlayer, w)), np.zeros((self.n_bgc, w))
        self._chk(self._f("get_tracer_output")(self._h, _dp(a), _dp(b)), "get_tracer_output")
        return a, b

    def ensemble_stats(self, names):
        """{name: Stat} over the columns without a STOP code; names from SCALARS or "N_active" (samsim_get_ensemble_stats)"""
        names = list(names)
        slots = (C.c_int32 * len(names))(*[-1 if n == "N_active" else S[n] for n in names])
        out = (Stat * len(names))()
        self._chk(self._f("get_ensemble_stats")(self._h, len(names), slots, out), "get_ensemble_stats")
        return {n: out[i] for i, n in enumerate(names)}

    def run_to_output(self) -> Output:
        """advance to (and through) the next output point of mo_grotz.f90:340 and return its snapshot"""
        self.step(self.steps_to_output())
        return self.get_output()

    def close(self):
        if self._h:
            self._f("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


HIP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsamsim_hip.so")
