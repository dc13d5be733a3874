"""The C-ABI shared library loads on a machine WITHOUT a GPU and exports every symbol include/samsim.h declares.
No compute call is made here."""
import ctypes as C
import os
import re

import pytest

import samsim_amd
from samsim_amd import testcases as tcs
from samsim_amd.capi import Config, HIP_LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "samsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(samsim_[a-z_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ["samsim_create", "samsim_set_forcing", "samsim_set_state", "samsim_step", "samsim_get_state",
                 "samsim_get_output", "samsim_get_status", "samsim_destroy"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(HIP_LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = samsim_amd.load()
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/samsim.h but not exported"
    assert lib.samsim_abi_version() == samsim_amd.capi.ABI_VERSION


def test_config_struct_layout_matches_header():
    """field order of samsim_config in the header == ctypes mirror"""
    text = open(os.path.join(ROOT, "include", "samsim.h")).read()
    body = text[text.index("typedef struct samsim_config {"):text.index("} samsim_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int32_t|double)\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [n for n, _ in Config._fields_]


def test_no_silent_cpu_fallback():
    """without a HIP device the product refuses to create a handle (it never computes on the CPU)"""
    lib = samsim_amd.load()
    lib.samsim_device_count.restype = C.c_int
    if lib.samsim_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg, _ = tcs.testcase1(1)
    with pytest.raises(samsim_amd.SamsimError) as e:
        samsim_amd.hip_solver(cfg, 4)
    assert e.value.code == -4


def test_unsupported_configurations_are_refused_before_any_device_call():
    """samsim_create validates the configuration first: flag values and testcases the path does not implement come back as
    SAMSIM_ERR_UNSUPPORTED (-2) with or without a GPU (a valid one gets as far as the device check)"""
    for flag, value in (("prescribe_flag", 3), ("flush_flag", 3), ("grav_flag", 4), ("testcase", 99), ("testcase", 103)):
        cfg, _ = tcs.testcase1(1)
        setattr(cfg, flag, value)
        with pytest.raises(samsim_amd.SamsimError) as e:
            samsim_amd.hip_solver(cfg, 4)
        assert e.value.code == -2, (flag, value)
    cfg, _ = tcs.testcase2(1)
    cfg.lab_snow_flag = 1
    with pytest.raises(samsim_amd.SamsimError) as e:
        samsim_amd.hip_solver(cfg, 4)
    assert e.value.code == -2


def test_handle_size_limit_is_checked_before_any_device_call():
    """one handle addresses its [38][ncol] scalar block with 32-bit byte offsets: 38 * ncol * 8 >= 4 GiB is SAMSIM_ERR_ARG (-1),
    whatever nlayer is (a thin column does not lift the bound: the layer block is not what limits a handle)"""
    from samsim_amd.capi import NSCAL
    max_ncol = ((1 << 32) - 1) // (8 * NSCAL)
    text = open(os.path.join(ROOT, "include", "samsim.h")).read()
    assert "#define SAMSIM_MAX_NCOL ((int64_t)(((1ull << 32) - 1) / (8ull * SAMSIM_NSCAL)))" in text
    for nlayer, n_top, n_bottom in ((100, 20, 20), (20, 5, 5)):
        cfg, _ = tcs.testcase4(1, nlayer=nlayer, n_top=n_top, n_bottom=n_bottom)
        with pytest.raises(samsim_amd.SamsimError) as e:
            samsim_amd.hip_solver(cfg, max_ncol + 1)
        assert e.value.code == -1, nlayer
        with pytest.raises(samsim_amd.SamsimError) as e:    # one column fewer passes the size check (and stops at the device check
            samsim_amd.hip_solver(cfg, 1 << 40)             # here, or at the allocation on a GPU box): not probed with a real size
        assert e.value.code == -1


def test_product_does_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "samsim_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "samsim_oracle" not in txt and "oracle_lib" not in txt, f
