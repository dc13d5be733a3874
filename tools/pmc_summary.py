#!/usr/bin/env python3
"""Summarise rocprofv3 outputs for samsim_step_kernel: kernel-trace stats plus per-launch medians of PMC counters collected in
separate passes (FETCH_SIZE / WRITE_SIZE / SQ_*), with the gfx950 HBM-byte correction of MI355X_MICROARCH.md
(hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; calibrated for this access width in profiles/r1_hbm_counter_calibration.txt).

    python3 tools/pmc_summary.py --dir gpurun_out/prof --ncol 1048576 --nlayer 80 --substeps 20 [--lib path.so] > profiles/rN_pmc_summary.json
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import statistics
import sys

csv.field_size_limit(1 << 30)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--ncol", type=int, default=1 << 20)
    ap.add_argument("--nlayer", type=int, default=80)
    ap.add_argument("--substeps", type=int, default=20)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--kernel", default="samsim_step_kernel")
    ap.add_argument("--parts", type=int, default=2,
                    help="launches of the kernel per step (a step of a large ensemble is two concurrent launches on two streams): the "
                         "counters of the parts are added, the step time is taken from the first start to the last end of its parts")
    a = ap.parse_args()
    out = {"kernel": a.kernel, "ncol": a.ncol, "nlayer": a.nlayer, "timesteps_per_launch": a.substeps}
    if a.lib and os.path.exists(a.lib):
        out["lib_md5"] = hashlib.md5(open(a.lib, "rb").read()).hexdigest()
    counters = {}
    for f in glob.glob(os.path.join(a.dir, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            per_dispatch = {}
            for r in csv.DictReader(fh):
                if a.kernel not in r["Kernel_Name"]:
                    continue
                per_dispatch.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per_dispatch[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
                # (rocprofv3's VGPR_Count column is not the compiler's register count -- 64 where the kernel is built for 128 --
                # so it is left out: profiles/r*_resource_usage.txt holds the compiler's remarks)
                out.setdefault("lds_block_bytes", int(r["LDS_Block_Size"]))
                out.setdefault("scratch_bytes_per_lane", int(r["Scratch_Size"]))
            for n, d in per_dispatch.items():
                ids = sorted(d, key=lambda k: int(k))
                steps = [sum(d[i] for i in ids[j:j + a.parts]) for j in range(0, len(ids) - a.parts + 1, a.parts)]
                counters[n] = {"median": statistics.median(steps), "launches": len(ids), "steps": len(steps)}
    out["launches_per_step"] = a.parts
    out["counters_per_launch"] = counters   # (per step: the sum over the launches of a step)
    for f in glob.glob(os.path.join(a.dir, "**", "*_kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if a.kernel in r["Name"]:
                    out["kernel_stats"] = {"calls": int(r["Calls"]), "average_ms": float(r["AverageNs"]) / 1e6,
                                           "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
    # one step = a.parts consecutive dispatches of the kernel: from the first start to the last end
    for f in glob.glob(os.path.join(a.dir, "**", "*_kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rows = [(int(r["Dispatch_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(fh)
                    if a.kernel in r["Kernel_Name"]]
        rows.sort()
        steps = [rows[j:j + a.parts] for j in range(0, len(rows) - a.parts + 1, a.parts)]
        if steps:
            dur = [(max(e for _, _, e in st) - min(b for _, b, _ in st)) / 1e6 for st in steps]
            span = (max(e for _, _, e in rows) - min(b for _, b, _ in rows)) / 1e6
            out["kernel_trace"] = {"steps": len(steps), "average_ms": sum(dur) / len(dur), "min_ms": min(dur), "max_ms": max(dur),
                                   "first_start_to_last_end_per_step_ms": span / len(steps)}
    cells = a.ncol * a.nlayer * a.substeps
    waves_cells = cells / 64.0
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        b = (2.0 * counters["FETCH_SIZE"]["median"] + counters["WRITE_SIZE"]["median"]) * 1024.0
        out["hbm_bytes_per_launch"] = b
        out["hbm_bytes_per_layer_cell"] = b / cells
        out["algorithmic_bytes_per_launch"] = 16.0 * (4 * a.nlayer + 24) * a.ncol * a.substeps
        out["traffic_over_algorithmic"] = b / out["algorithmic_bytes_per_launch"]
        if "kernel_trace" in out:
            out["hbm_TBps_at_traced_duration"] = b / (out["kernel_trace"]["average_ms"] * 1e-3) / 1e12
    per_cell = {}
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_FLAT", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD",
              "SQ_INSTS_VMEM_WR"):
        if n in counters:
            per_cell[n] = counters[n]["median"] / waves_cells
    if per_cell:
        out["instructions_per_layer_cell_of_a_wave"] = per_cell
    if "SQ_ACTIVE_INST_VALU" in counters and "GRBM_GUI_ACTIVE" in counters:
        cyc = counters["GRBM_GUI_ACTIVE"]["median"] / 8.0
        out["gpu_cycles_per_launch"] = cyc
        out["valu_busy_frac"] = counters["SQ_ACTIVE_INST_VALU"]["median"] * 4.0 / (1024.0 * cyc)
    if "SQ_WAVE_CYCLES" in counters:
        wc = counters["SQ_WAVE_CYCLES"]["median"]
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in counters:
                out[n + "_frac_of_wave_cycles"] = counters[n]["median"] / wc
        if "GRBM_GUI_ACTIVE" in counters:
            out["mean_waves_per_simd"] = wc * 4.0 / (1024.0 * counters["GRBM_GUI_ACTIVE"]["median"] / 8.0)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
