#!/usr/bin/env python3
"""Fill the @MARKER@ fields of DESIGN.md (the table and the paragraph of section 4 carry them while a round is in progress) from the
committed summaries under profiles/; run after tools/collect_profiles.py.  Once filled, the markers are gone: put them back to refill."""
import json
import re
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r3"
P = "profiles/%s_" % rnd
b = json.load(open(P + "bench.json"))
pm = json.load(open(P + "pmc_summary_substeps500.json"))
tc1 = json.load(open(P + "bench_tc1.json"))
cfg5 = json.load(open(P + "bench_cfg5.json"))
ins = pm["instructions_per_layer_cell_of_a_wave"]
ms = b["roofline"]["mean_launch_ms"]
v = {
    "MS": "%.0f" % ms, "CTS": "%.2e" % b["value"], "FRAC": "%.3f" % b["roofline"]["frac"],
    "BPC": "%.0f" % pm["hbm_bytes_per_layer_cell"], "RATIO": "%.2f" % pm["traffic_over_algorithmic"],
    "VALU": "%.0f" % ins["SQ_INSTS_VALU"], "SALU": "%.0f" % ins["SQ_INSTS_SALU"], "BRANCH": "%.0f" % ins["SQ_INSTS_BRANCH"],
    "SCR": "%d" % pm["scratch_bytes_per_lane"],
    "MELT": "%.2e" % b["extra"]["stages"]["day360"]["column_timesteps_per_s"],
    "M345": "%.2e" % b["extra"]["stages"]["day345"]["column_timesteps_per_s"],
    "TRACEMS": "%.0f" % pm["kernel_trace"]["first_start_to_last_end_per_step_ms"],
    "F300": "%.2e" % b["extra"]["first_300_days"]["column_timesteps_per_s"],
    "TC1": "%.3f" % tc1["roofline"]["frac"], "CFG5": "%.3f" % cfg5["roofline"]["frac"],
    "VBUSY": "%.0f" % (100 * pm["valu_busy_frac"]),
    "TBS": "%.1f" % (pm["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e12),
}
left = []
for doc in ("DESIGN.md", "README.md"):
    tmpl = open(doc).read()
    for k, x in v.items():
        tmpl = tmpl.replace("@%s@" % k, x)
    left += re.findall(r"@[A-Z0-9]+@", tmpl)
    open(doc, "w").write(tmpl)
print(v, "unfilled:", left)
