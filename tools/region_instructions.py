#!/usr/bin/env python3
"""Per-region instruction table of samsim_step_kernel<KSheba> (profiles/r3_region_instructions.json).

  static   census of the regions between the ISA_MARK comments of a `hipcc -S -DSAMSIM_ISA_MARKS=1` listing of the product source
           (vector / scalar / branch / memory instructions as written, rare branches included: an upper bound of what a trip runs)
  trips    how often a region runs per layer-cell of a wave, from the event counters of the -DSAMSIM_STAMPS=2 build
           (tools/stamps.py: Newton evaluations per cell at the wave maximum, loop trips, draining rows)
  measured totals per layer-cell of a wave from the rocprofv3 PMC passes (tools/pmc_summary.py): the whole kernel, exact

usage: region_instructions.py listing.s counters.json pmc_summary.json [stamps.json] > profiles/r3_region_instructions.json
"""
import collections
import json
import re
import sys


def classify(op):
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "global_load"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "global_store"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane_moves"
    if op.startswith("v_mov"): return "valu_moves"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    listing, counters, pmc = sys.argv[1], json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    stamps = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
    lines = open(listing).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and "6KShebaE" in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    marks = [(i, l.split("ISA_MARK")[1].strip()) for i, l in enumerate(body) if "ISA_MARK" in l]

    def census(a, b):
        c = collections.Counter()
        for l in body[a:b]:
            m = re.match(r"^\s+([a-z_0-9]+)", l)
            if not m or l.strip().startswith(";") or l.strip().startswith("."):
                continue
            c[classify(m.group(1))] += 1
        c["vector_total"] = c["valu"] + c["valu_moves"] + c["lane_moves"]
        return dict(c)

    regions = {}
    for (i, a), (j, b) in zip(marks, marks[1:]):
        key = f"{a}..{b}"
        if key not in regions:                 # (the up sweep's body is instantiated once per thickness variant: first = regular waves)
            regions[key] = census(i, j)
    evals = counters.get("newton_evals_per_cell_wave_max")
    out = {
        "what": "samsim_step_kernel<KSheba>, product source with -DSAMSIM_ISA_MARKS=1; see tools/region_instructions.py",
        "static_census_between_marks": regions,
        "whole_kernel_static": census(0, len(body)),
        "trip_counts": {
            "newton_evaluations_per_layer_cell_wave_max": evals,
            "newton_evaluations_per_layer_cell_lane_mean": counters.get("newton_evals_per_cell_lane_mean"),
            "up_sweep_trips": counters.get("up_trips"), "down_sweep_layer_trips": counters.get("down_trips"),
            "wave_steps": counters.get("wave_steps"),
            "rows_with_a_draining_column": counters.get("drain_wave"),
            "interior_rows_of_the_down_sweep": counters.get("down_rows_interior"),
            "of_them_without_brine_expulsion_in_any_column": counters.get("down_rows_without_expulsion"),
            "full_first_sweeps": counters.get("dirty"),
        },
        "measured_per_layer_cell_of_a_wave": pmc.get("instructions_per_layer_cell_of_a_wave"),
        "measured_hbm_bytes_per_layer_cell": pmc.get("hbm_bytes_per_layer_cell"),
        "measured_valu_busy_frac": pmc.get("valu_busy_frac"),
        "lib_md5_of_the_pmc_run": pmc.get("lib_md5"),
    }
    if stamps:
        out["time_shares_s_memtime_build"] = stamps.get("share")
        out["time_shares_note"] = ("-DSAMSIM_STAMPS=1 build: every stamp is an s_memtime + wait (~100-200 cycles), which inflates the "
                                   "short regions (loop back-edges: t_up, t_down_fused)")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
