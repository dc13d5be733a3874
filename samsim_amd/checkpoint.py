"""Binary checkpoint / restart files for a resident ensemble (the reference has none: a
run there always starts from `init(testcase)`).

The file is a little-endian stream that the Fortran host reads and writes as well (host/host_driver.f90, ACCESS='stream'):

    header   int64 magic "SAMCHK02", ncol, nlayer, narr, nscal, testcase,
             float64 time, int64 step, n_time_out, time_counter, n_outputs, int64 n_bgc, int64 has_status, int64 reserved[3]
    chunks   int64 col0, ncols, then lay[narr][nlayer][ncols], scal[nscal][ncols] (float64), n_active[ncols] (int32),
             status[ncols], err_layer[ncols] (int32) and err_step[ncols] (int64): the STOP code of a frozen column and where
             it failed (samsim_get_status; has_status = 1 in every "SAMCHK02" file -- the chunk layout without them was
             "SAMCHK01", which this version refuses by its magic instead of mis-reading it),
             and with tracers (n_bgc > 0) bgc_abs[n_bgc][nlayer][ncols], bgc_bottom[n_bgc][ncols] (float64)
             ... until ncol columns are covered

It holds what `samsim_get_state` / `samsim_get_clock` return: with all 15 layer arrays (`narr = NARR`, the default) a
restart continues bit for bit; with the 4 prognostic arrays (`narr = NPROG`) the diagnostics are rebuilt by the first
sweep of the next step and only the paths that read last step's temperature (rain into open water) see a difference.
The number of tracers, the tank totals and the forcing are configuration: the caller sets them (`set_tracers`,
`set_forcing`) before `load`, which restores the per-column tracer amounts and the concentration of the water below.
"""
from __future__ import annotations

import struct

import numpy as np

from .capi import NARR, NPROG, NSCAL, Solver, State

MAGIC = int.from_bytes(b"SAMCHK02", "little")
MAGIC_OLD = int.from_bytes(b"SAMCHK01", "little")   # chunks without the status block: refused
_HDR = struct.Struct("<6q d 4q 5q")


def save(solver: Solver, path: str, narr: int = NARR, chunk: int = 65536) -> None:
    """stream the state of `solver` to `path`, `chunk` columns at a time"""
    assert narr in (NARR, NPROG)
    k = solver.get_clock()
    n_bgc = int(getattr(solver, "n_bgc", 0)) if int(solver.cfg.bgc_flag) == 2 else 0
    status, err_step, err_layer = solver.get_status()
    with open(path, "wb") as f:
        f.write(_HDR.pack(MAGIC, solver.ncol, solver.nlayer, narr, NSCAL, int(solver.cfg.testcase), float(k.time),
                          int(k.step), int(k.n_time_out), int(k.time_counter), int(k.n_outputs), n_bgc, 1, 0, 0, 0))
        c0 = 0
        while c0 < solver.ncol:
            n = min(chunk, solver.ncol - c0)
            st = solver.get_state(col0=c0, ncols=n, narr=narr)
            f.write(struct.pack("<2q", c0, n))
            f.write(np.ascontiguousarray(st.lay, dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(st.scal, dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(st.n_active, dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(status[c0:c0 + n], dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(err_layer[c0:c0 + n], dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(err_step[c0:c0 + n], dtype="<i8").tobytes())
            if n_bgc:
                a, b = solver.get_tracer_state(c0, n)
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
                f.write(np.ascontiguousarray(b, dtype="<f8").tobytes())
            c0 += n


def read_header(path: str) -> dict:
    with open(path, "rb") as f:
        v = _HDR.unpack(f.read(_HDR.size))
    if v[0] == MAGIC_OLD:
        raise ValueError(f"{path}: checkpoint layout SAMCHK01 (no status block per chunk) is not read by this version")
    if v[0] != MAGIC:
        raise ValueError(f"{path}: not a SAMSIM checkpoint")
    return dict(ncol=v[1], nlayer=v[2], narr=v[3], nscal=v[4], testcase=v[5], time=v[6], step=v[7], n_time_out=v[8],
                time_counter=v[9], n_outputs=v[10], n_bgc=v[11], has_status=v[12])


def load(solver: Solver, path: str) -> dict:
    """upload the state and the clock stored in `path` into `solver` (same ncol and nlayer); returns the header"""
    h = read_header(path)
    if (h["ncol"], h["nlayer"]) != (solver.ncol, solver.nlayer) or h["nscal"] != NSCAL or h["narr"] not in (NARR, NPROG):
        raise ValueError(f"{path}: holds {h['ncol']} columns x {h['nlayer']} layers x {h['nscal']} scalars, "
                         f"the solver {solver.ncol} x {solver.nlayer} x {NSCAL}")
    n_bgc = h["n_bgc"]
    if n_bgc != (int(getattr(solver, "n_bgc", 0)) if int(solver.cfg.bgc_flag) == 2 else 0):
        raise ValueError(f"{path}: holds {n_bgc} tracers, the solver {getattr(solver, 'n_bgc', 0)} (call set_tracers first)")
    with open(path, "rb") as f:
        f.seek(_HDR.size)
        done = 0
        while done < h["ncol"]:
            c0, n = struct.unpack("<2q", f.read(16))
            lay = np.frombuffer(f.read(8 * h["narr"] * h["nlayer"] * n), dtype="<f8").reshape(h["narr"], h["nlayer"], n)
            scal = np.frombuffer(f.read(8 * NSCAL * n), dtype="<f8").reshape(NSCAL, n)
            na = np.frombuffer(f.read(4 * n), dtype="<i4")
            solver.set_state(State(np.ascontiguousarray(lay, dtype=np.float64), np.ascontiguousarray(scal, dtype=np.float64),
                                   np.ascontiguousarray(na, dtype=np.int32)), c0)
            if h["has_status"]:   # set_state cleared the STOP codes of these columns: frozen columns stay frozen
                status = np.frombuffer(f.read(4 * n), dtype="<i4")
                err_layer = np.frombuffer(f.read(4 * n), dtype="<i4")
                err_step = np.frombuffer(f.read(8 * n), dtype="<i8")
                solver.set_status(status, err_step, err_layer, c0)
            if n_bgc:
                a = np.frombuffer(f.read(8 * n_bgc * h["nlayer"] * n), dtype="<f8").reshape(n_bgc, h["nlayer"], n)
                b = np.frombuffer(f.read(8 * n_bgc * n), dtype="<f8").reshape(n_bgc, n)
                solver.set_tracer_state(np.ascontiguousarray(a, dtype=np.float64), c0)
                solver.set_tracer_bottom(np.ascontiguousarray(b, dtype=np.float64), c0)
            done += n
    solver.set_clock(time=h["time"], step=h["step"], n_time_out=h["n_time_out"], time_counter=h["time_counter"],
                     n_outputs=h["n_outputs"])
    return h
