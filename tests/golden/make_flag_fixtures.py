#!/usr/bin/env python3
"""Fixture generator (build container only; needs oracle/_ref): the flag values no shipped testcase uses --
harmonic_flag 1, freeboard_snow_flag 1, snow_flush_flag 0, bottom_flag 2 -- pinned on the reference itself: testcase 4 /
SHEBA run by the flang-built reference with ONE flag overridden after init (SAMSIM_REF_* of oracle/ref_hook), scalars at
every output day through the first melt season, per-layer state at selected days.

    tests/golden/tc4_flag_<name>_ref.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import RUN, pack, run_ref  # noqa: E402
from tests.refdump import read_dump  # noqa: E402

VARIANTS = {"harmonic1": {"SAMSIM_REF_HARMONIC": "1"}, "freeboard_snow1": {"SAMSIM_REF_FREEBOARD_SNOW": "1"},
            "snow_flush0": {"SAMSIM_REF_SNOW_FLUSH": "0"}, "bottom2": {"SAMSIM_REF_BOTTOM": "2"}}
DAYS_LAYERS = [1, 10, 40, 67, 100, 150, 200, 250, 300, 330, 345, 350, 355, 360, 370, 380, 390, 400, 420]
TF_PAIRS = [66, 70, 120, 250, 345, 352, 358, 365, 380, 400]     # teacher-forcing pairs (day D, day D+1), full mid-step state
NDAYS = 425


def main():
    names = sys.argv[1:] or list(VARIANTS)
    for name in names:
        dump = os.path.join(RUN, f"tc4_flag_{name}.bin")
        recs = read_dump(dump) if os.path.exists(dump) and os.environ.get("REPACK") else \
            run_ref(4, f"tc4_flag_{name}.bin", dict(VARIANTS[name], SAMSIM_REF_MAXSTEPS=str(8641 * NDAYS)))
        d = pack([recs[x - 1] for x in DAYS_LAYERS if x - 1 < len(recs)])
        d["day_index"] = np.array([x for x in DAYS_LAYERS if x - 1 < len(recs)])
        for k, v in pack(recs, with_layers=False).items():
            d["all_" + k] = v
        for k, v in pack([recs[i - 1] for q in TF_PAIRS for i in (q, q + 1)]).items():
            d["tf_" + k] = v
        d["tf_days"] = np.array(TF_PAIRS)
        d["flag_env"] = np.array(list(VARIANTS[name].items())[0])
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"tc4_flag_{name}_ref.npz"), **d)
        print(name, "records", len(recs), flush=True)


if __name__ == "__main__":
    main()
