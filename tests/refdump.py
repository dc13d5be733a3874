"""TEST INFRASTRUCTURE: reader for the binary records written by oracle/ref_hook/ref_output_hook.f90
(full-precision dumps of the flang-built reference, oracle/_ref/samsim_ref_dump)."""
import numpy as np

MAGIC = 0x53414D53
NSCAL_REF = 48
REF_SCALARS = ["time", "dt", "thick_0", "T_bottom", "S_bu_bottom", "T_top", "T2m", "fl_q_bottom", "m_snow",
               "H_abs_snow", "S_abs_snow", "thick_snow", "psi_s_snow", "psi_l_snow", "psi_g_snow", "T_snow",
               "phi_s", "liquid_precip", "solid_precip", "fl_Q_snow", "melt_thick", "melt_thick_snow",
               "melt_out1", "melt_out2", "melt_out3", "freeboard", "T_freeze", "albedo", "fl_sw", "fl_lw",
               "fl_rest", "grav_drain", "grav_salt", "grav_temp", "melt_err", "energy_stored", "freshwater",
               "total_resist", "thickness", "bulk_salin", "fl_Q1", "fl_Q_bottom", "thick_min", "n_time_out",
               "i_time_out", "i_time"]
REF_ARRAYS = ["H_abs", "S_abs", "m", "thick", "T", "phi", "psi_s", "psi_l", "psi_g", "S_bu", "S_br", "V_ex",
              "ray", "perm", "flush_v", "flush_h", "fl_rad", "fl_Q"]


def read_dump(path):
    """returns a list of dicts: kind, step, N_active, Nlayer, time_counter, scal{name:val}, arr{name:array}"""
    buf = np.fromfile(path, dtype=np.uint8)
    recs, pos = [], 0
    while pos < len(buf):
        hdr = buf[pos:pos + 32].view(np.int32)
        assert hdr[0] == MAGIC, "bad magic in reference dump"
        kind, step, na, nl, tc = (int(x) for x in hdr[1:6])
        pos += 32
        sc = buf[pos:pos + 8 * NSCAL_REF].view(np.float64)
        pos += 8 * NSCAL_REF
        arrs = {}
        for name in REF_ARRAYS:
            arrs[name] = buf[pos:pos + 8 * nl].view(np.float64).copy()
            pos += 8 * nl
        recs.append(dict(kind=kind, step=step, N_active=na, Nlayer=nl, time_counter=tc,
                         scal={n: float(sc[i]) for i, n in enumerate(REF_SCALARS)}, arr=arrs))
    return recs
