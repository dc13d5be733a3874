#!/usr/bin/env python3
"""Copy the summaries of a profiling run (gpurun_out/<dir>, written by a gpurun_d*.sh script: tools/profile.sh + tools/pmc_summary.py +
tools/stamps.py + bench.py) into profiles/ under the round's names and refresh profiles/traffic.json (HBM bytes per step, keyed by
workload and by the md5 of the profiled library).

    python tools/collect_profiles.py gpurun_out/d3 r2
"""
import glob
import json
import os
import shutil
import sys

src, rnd = sys.argv[1], sys.argv[2]
P = "profiles"
pairs = [("bench_default.json", f"{rnd}_bench.json"), ("pmc_summary_500.json", f"{rnd}_pmc_summary_substeps500.json"),
         ("prof500/bench_under_rocprof.json", f"{rnd}_bench_under_rocprofv3_substeps500.json"),
         ("stamps_winter.json", f"{rnd}_stamps_winter.json"), ("stamps_melt.json", f"{rnd}_stamps_melt.json"),
         ("counters_winter.json", f"{rnd}_counters_winter.json"), ("counters_melt.json", f"{rnd}_counters_melt.json"),
         ("bench_tc1.json", f"{rnd}_bench_tc1.json"), ("bench_cfg5.json", f"{rnd}_bench_cfg5.json"),
         ("bench_nlayer100.json", f"{rnd}_bench_nlayer100.json"), ("bench_two_ranks_one_gpu.json", f"{rnd}_bench_two_ranks_one_gpu.json"),
         ("path_equiv_cfg5.json", f"{rnd}_path_equiv_cfg5.json"),
         ("path_equiv_sheba_ensemble_80.npz.json", f"{rnd}_path_equiv_sheba_ensemble_80.json"),
         ("path_equiv_sheba_ensemble_80_day345.npz.json", f"{rnd}_path_equiv_sheba_ensemble_80_day345.json"),
         ("div_probe.txt", f"{rnd}_div_probe.txt"), ("layout_probe.txt", f"{rnd}_layout_probe.txt"),
         ("bench_steps20.json", f"{rnd}_bench_steps20.json"), ("melt_ensemble_status.json", f"{rnd}_melt_ensemble_status.json")]
for a, b in pairs:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(P, b))
for f in glob.glob(os.path.join(src, "prof500/trace/*/*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(P, f"{rnd}_kernel_stats_substeps500.csv"))
for f in glob.glob(os.path.join(src, "prof500/trace/*/*_kernel_trace.csv")):
    # the step kernel's dispatches only (start / end of the two launches of every step)
    with open(f) as fh, open(os.path.join(P, f"{rnd}_kernel_trace_substeps500.csv"), "w") as out:
        for i, line in enumerate(fh):
            if i == 0 or "samsim_step_kernel" in line:
                out.write(line)
d = json.load(open(os.path.join(src, "pmc_summary_500.json")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
# (src_md5: the sources in the tree NOW -- run this right after the profiling call, before the sources change)
t = {"sheba:1048576:80:500": {"bytes_per_launch": d["hbm_bytes_per_launch"], "lib_md5": d["lib_md5"], "src_md5": bench.source_md5(),
                                "source": f"profiles/{rnd}_pmc_summary_substeps500.json",
                                "note": "per step of 500 time steps = the two launches of the step added (2*FETCH_SIZE + WRITE_SIZE)*1024"}}
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps(t, indent=1))
