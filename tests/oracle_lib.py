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
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(ORACLE_SO)
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


def oracle_solver(cfg, ncol) -> Solver:
    return Solver(load_oracle(), "oracle_", cfg, ncol)
