#!/usr/bin/env python3
"""Fixtures for the remaining ERA-interim sites of the reference (input/ERA-interim/{75N180E,80N00E,75N00W,85N180E,80N90E}-p2;
SURVEY.md section 8 f.4): their forcing tables, and the reference itself (oracle/_ref/samsim_ref_dump, the flang build of the
unmodified physics) run on testcase 4 in a directory that holds each site's tables -- first 150 output days, full precision.

    python tests/golden/make_site_fixtures.py      (build container only: needs /root/reference and oracle/_ref)

writes tests/golden/era_sites_forcing_more.npz and tests/golden/tc4_sites_ref.npz
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from samsim_amd import testcases as tcs  # noqa: E402
from tests.refdump import read_dump  # noqa: E402
from tests.golden.make_golden import pack, REF, OUT  # noqa: E402

SITES = ["75N180E", "80N00E", "75N00W", "85N180E", "80N90E"]
DAYS = 150


def main():
    forcing, ref = {}, {}
    for site in SITES:
        src = os.path.join(REF, "input", "ERA-interim", site + "-p2")
        for n, a in zip(("fl_sw", "fl_lw", "T2m", "precip"), tcs.read_forcing(src)):
            forcing[f"{site}_{n}"] = a
        run = os.path.join(ROOT, "oracle", "_ref", "run_" + site)
        os.makedirs(os.path.join(run, "output"), exist_ok=True)
        for n in ("flux_lw", "flux_sw", "T2m", "precip"):
            dst = os.path.join(run, n + ".txt.input")
            if not os.path.lexists(dst):
                os.symlink(os.path.join(src, n + ".txt.input"), dst)
        dump = os.path.join(run, "tc4_site.bin")
        if not os.path.exists(dump):
            subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "samsim_ref_dump"), "4"], cwd=run, stdout=subprocess.DEVNULL,
                                  env=dict(os.environ, SAMSIM_REF_DUMP="tc4_site.bin", SAMSIM_REF_MAXSTEPS=str(8641 * DAYS + 1)))
        recs = read_dump(dump)[:DAYS]
        sel = [0, 50, 100, DAYS - 1]
        for k, v in pack([recs[i] for i in sel]).items():
            ref[f"{site}_{k}"] = v
        ref[f"{site}_index"] = np.array(sel)
        for k, v in pack(recs, with_layers=False).items():
            ref[f"{site}_all_{k}"] = v
        print(site, len(recs), "records, N_active", int(ref[f"{site}_all_N_active"][-1]))
    np.savez_compressed(os.path.join(OUT, "era_sites_forcing_more.npz"), **forcing)
    np.savez_compressed(os.path.join(OUT, "tc4_sites_ref.npz"), **ref)


if __name__ == "__main__":
    main()
