"""samsim_amd -- MI355X-native batched sea-ice column solver (drop-in for SAMSIM's time-loop body).

The compute path is the HIP library samsim_amd/csrc/libsamsim_hip.so behind the C-ABI of include/samsim.h.
There is no CPU fallback: `load()` raises when the library has not been built.
"""
from __future__ import annotations

import ctypes as _C
import os as _os

from . import capi, checkpoint, testcases  # noqa: F401
from .capi import Config, State, Output, Solver, SamsimError, HIP_LIB_PATH  # noqa: F401

_lib = None


def load() -> _C.CDLL:
    """dlopen the HIP product library (fails loudly if it is not built)."""
    global _lib
    if _lib is None:
        path = _os.environ.get("SAMSIM_HIP_LIB", HIP_LIB_PATH)  # tuning variants of the same HIP library
        if not _os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _lib = _C.CDLL(path)
        _lib.samsim_abi_version.restype = _C.c_int
        if _lib.samsim_abi_version() != capi.ABI_VERSION:
            raise RuntimeError("libsamsim_hip.so ABI version mismatch")
    return _lib


def hip_solver(cfg: Config, ncol: int, device: int = 0) -> Solver:
    """samsim_create on `device`.  SAMSIM_TEST_SPLIT_BLOCKS (read HERE, by the Python mirror -- the library reads no environment)
    lowers the block count from which a step runs as two concurrent launches: the whole GPU suite can be run with every launch
    split (samsim_set_launch_split)."""
    s = Solver(load(), "samsim_", cfg, ncol, device)
    if "SAMSIM_TEST_SPLIT_BLOCKS" in _os.environ:
        s.set_launch_split(int(_os.environ["SAMSIM_TEST_SPLIT_BLOCKS"]), 4)
    return s
