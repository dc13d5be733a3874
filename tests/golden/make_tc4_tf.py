#!/usr/bin/env python3
"""Fixture generator (build container only; needs oracle/_ref, see oracle/build_ref.sh): teacher-forcing pairs of the
REFERENCE's own testcase-4 / SHEBA run through open water, freeze-up and the growth season -- the full mid-step state the
reference dumps at output day D and at day D+1 -- for days sampled over 0-335:

    tests/golden/tc4_tf_growth_ref.npz    (same keys as the tf_* block of tc4_ref_fullprec.npz, which holds the melt-season pairs)

tests/test_gpu_reference_windows.py starts the HIP path from day D and requires the reference's day D+1 record."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import pack, run_ref  # noqa: E402

PAIRS = [2, 20, 45, 66, 100, 150, 200, 250, 300, 330]


def main():
    recs = run_ref(4, "tc4_growth.bin", {"SAMSIM_REF_MAXSTEPS": str(8641 * 333)})
    d = {}
    tf = pack([recs[i - 1] for p in PAIRS for i in (p, p + 1)])
    for k, v in tf.items():
        d["tf_" + k] = v
    d["tf_days"] = np.array(PAIRS)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tc4_tf_growth_ref.npz"), **d)
    print("records", len(recs), "pairs", PAIRS)


if __name__ == "__main__":
    main()
