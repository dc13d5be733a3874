import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """the two tests that wait for the background free runs (tests/background_runs.py) go last: the runs then have the whole session
    beside them, and the session does not sit waiting for them in the middle"""
    waits = ("test_sheba_free_run_from_open_water_against_the_reference_records",
             "test_the_five_later_era_sites_free_run_against_the_reference_records")
    last = [i for i in items if i.name in waits]
    if last:
        items[:] = [i for i in items if i.name not in waits] + last


@pytest.fixture(scope="session")
def oracle_lib():
    from tests.oracle_lib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session", autouse=True)
def _background_free_runs(request):
    """the two long single-wave free runs of the GPU suite start with the session's first GPU test and run beside the others
    (tests/background_runs.py); nothing happens in a CPU session"""
    gpu_session = any(item.get_closest_marker("gpu") is not None for item in request.session.items)
    started = False
    if gpu_session:
        try:
            import ctypes
            import samsim_amd
            lib = samsim_amd.load()
            lib.samsim_device_count.restype = ctypes.c_int
            if lib.samsim_device_count() > 0 and len([i for i in request.session.items if i.get_closest_marker("gpu")]) > 10:
                from tests import background_runs
                background_runs.start_all()
                started = True
        except Exception:      # noqa: BLE001  (a missing library is the GPU tests' own failure to report)
            started = False
    yield
    if started:
        from tests import background_runs
        background_runs.stop_all()
