"""TEST INFRASTRUCTURE: loads the CPU oracle (oracle/liboracle.so) and wraps it with the same Solver class
that drives the HIP library, so parity tests feed both identical inputs."""
import ctypes as C
import os
import subprocess

from samsim_amd.capi import Solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_lib = None


def load_oracle() -> C.CDLL:
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "samsim_oracle.c")
        so = os.environ.get("SAMSIM_ORACLE_SO", ORACLE_SO)     # the sanitizer build, tests/test_oracle_sanitized.py
        if so == ORACLE_SO and (not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(so)
        d = C.c_double
        _lib.oracle_func_S_br.restype = d
        _lib.oracle_func_S_br.argtypes = [C.c_int, d, d, C.c_int]
        _lib.oracle_func_ddT_S_br.restype = d
        _lib.oracle_func_ddT_S_br.argtypes = [C.c_int, d]
        _lib.oracle_func_T_freeze.restype = d
        _lib.oracle_func_T_freeze.argtypes = [d, C.c_int]
        _lib.oracle_func_density.restype = d
        _lib.oracle_func_density.argtypes = [d, d]
        _lib.oracle_func_albedo.restype = d
        _lib.oracle_func_albedo.argtypes = [d, d, d, d, C.c_int]
        _lib.oracle_func_k_snow.restype = d
        _lib.oracle_func_k_snow.argtypes = [d, d]
        _lib.oracle_getT.restype = None
        _lib.oracle_getT.argtypes = [C.c_int, d, d, d, C.POINTER(d), C.POINTER(d), C.POINTER(C.c_int)]
    return _lib


class OracleSolver(Solver):
    """the CPU oracle behind the same Python face as the HIP solver (oracle_* mirrors samsim_*, oracle/samsim_oracle.h)"""

    def _create(self, device):
        f = self._f("create")
        f.argtypes, f.restype = [C.POINTER(type(self.cfg)), C.c_int64, C.POINTER(C.c_void_p)], C.c_int
        return f(C.byref(self.cfg), C.c_int64(self.ncol), C.byref(self._h))

    def _bind_extra(self):
        f = self._f("step_part_b")
        f.argtypes, f.restype = [C.c_void_p], C.c_int
        f = self._f("set_threads")
        f.argtypes, f.restype = [C.c_void_p, C.c_int], None

    def synchronize(self):
        pass

    def step_part_b(self):
        self._chk(self._f("step_part_b")(self._h), "step_part_b")

    def set_threads(self, n):
        self._f("set_threads")(self._h, n)


def oracle_solver(cfg, ncol) -> OracleSolver:
    return OracleSolver(load_oracle(), "oracle_", cfg, ncol)
