"""Long single-wave free runs of the GPU suite, started once per session in background threads.

A free run of ONE column for hundreds of days is one wavefront on an otherwise idle GPU (about 0.14 ms per time step, latency-bound):
minutes of wall time during which the card has room for everything else.  The two such runs of the suite -- the unperturbed SHEBA
column against the reference's records to day 300 (tests/test_gpu_reference_windows.py) and the five later ERA-interim sites to day
150 (tests/test_gpu_secondary.py) -- are therefore started when the session's first GPU test begins, each in its own thread with its
own handle (the C-ABI is safe for one handle per thread; ctypes releases the interpreter lock during the calls), and run beside the
other tests.  Each thread only RECORDS the snapshot of every output day; the test that owns the run joins it and does all the
comparing, so a failure is reported by that test.
"""
import threading

import numpy as np

# A free-running column's output interval (8 641 steps) is 1.2 s of one kernel.  Calls of the other tests that wait for the whole
# device (hipMalloc / hipFree inside the library's set-up calls) would wait for it each time, so the background runs advance in
# launches of CHUNK steps and wait for each before enqueueing the next (launch granularity does not change a bit:
# tests/test_gpu_parity.py::test_launch_granularity_does_not_change_results).
CHUNK = 400

_lock = threading.Lock()
_runs = {}
_stop = threading.Event()


def _run_to_output(g):
    n = g.steps_to_output()
    while n > 0:
        m = min(n, CHUNK)
        g.step(m)
        g.synchronize()
        n -= m
    return g.get_output()


class Run:
    def __init__(self, name, fn):
        self.name, self.outputs, self.error, self.status = name, [], None, None
        self.thread = threading.Thread(target=self._main, args=(fn,), name=name, daemon=True)

    def _main(self, fn):
        try:
            fn(self)
        except BaseException as e:   # noqa: BLE001  (handed to the owning test)
            self.error = e

    def result(self, timeout=900):
        self.thread.join(timeout)
        assert not self.thread.is_alive(), f"{self.name}: still running after {timeout} s"
        if self.error is not None:
            raise self.error
        return self.outputs


def _sheba_free_run(run, days=301):
    import samsim_amd
    from samsim_amd import testcases as tcs
    from tests.helpers import sheba_forcing
    cfg, st = tcs.testcase4(1)
    g = samsim_amd.hip_solver(cfg, 1)
    g.set_forcing(*sheba_forcing())
    g.set_state(st)
    g.set_clock()
    g.set_output_window(0, 1)
    for _ in range(days):
        if _stop.is_set():
            break
        run.outputs.append(_run_to_output(g))
    run.status = g.get_status()[0].copy()
    g.close()


SITES = ["75N180E", "80N00E", "75N00W", "85N180E", "80N90E"]


def _sites_free_run(run, days=150):
    import samsim_amd
    from samsim_amd import testcases as tcs
    from tests.helpers import golden
    zm = golden("era_sites_forcing_more.npz")
    tables = [np.stack([zm[f"{s}_{n}"] for s in SITES]) for n in ("fl_sw", "fl_lw", "T2m", "precip")]
    ncol = len(SITES)
    cfg, st = tcs.testcase4(ncol)
    g = samsim_amd.hip_solver(cfg, ncol)
    g.set_forcing_sites(*tables, np.arange(ncol, dtype=np.int32), None, None)
    g.set_state(st)
    g.set_clock()
    g.set_output_window(0, ncol)
    for _ in range(days):
        if _stop.is_set():
            break
        run.outputs.append(_run_to_output(g))
    run.status = g.get_status()[0].copy()
    g.close()


_FUNCS = {"sheba_free_run": _sheba_free_run, "sites_free_run": _sites_free_run}


def start_all():
    """idempotent: called by the session fixture when the first GPU test is set up"""
    with _lock:
        for name, fn in _FUNCS.items():
            if name not in _runs:
                _runs[name] = Run(name, fn)
                _runs[name].thread.start()


def get(name):
    """the run `name` (started now if the session fixture did not: a test selected on its own)"""
    with _lock:
        if name not in _runs:
            _runs[name] = Run(name, _FUNCS[name])
            _runs[name].thread.start()
        return _runs[name]


def stop_all():
    _stop.set()
    with _lock:
        runs = list(_runs.values())
    for r in runs:
        r.thread.join(30)
