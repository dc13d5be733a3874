"""Rank process of tests/test_gpu_sharding.py: steps its shard [col0, col0+n) of the perturbed SHEBA ensemble through the HIP
library on the given device and writes the result.  Started as a fresh process (it is the first in its process tree to touch
the GPU), exactly like a rank of bench.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ensemble_shard(col0, n, nsteps, device=0, launches=1, split_blocks=None):
    import samsim_amd
    from samsim_amd import testcases as tcs
    from tests.helpers import load_checkpoint, sheba_forcing
    cfg, _ = tcs.testcase4(1)
    st1, clock = load_checkpoint("tc4_spunup_state.npz")
    g = samsim_amd.hip_solver(cfg, n, device=device)
    dT, ps = tcs.ensemble_perturbation(n, col0)      # counter-based: a function of the global column index
    g.set_forcing(*sheba_forcing(), dT, ps)
    g.set_state(st1.replicate(n))
    g.set_clock(**clock)
    if split_blocks is not None:
        g.set_launch_split(split_blocks, 4)
    for _ in range(launches):
        g.step(nsteps)
    st, status = g.get_state(), g.get_status()[0]
    g.close()
    return st, status


if __name__ == "__main__":
    rank, world, ncol, nsteps, device, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    from samsim_amd.shard import shard_range
    col0, n = shard_range(ncol, rank, world)
    st, status = ensemble_shard(col0, n, nsteps, device)
    np.savez(out, lay=st.lay, scal=st.scal, n_active=st.n_active, status=status, col0=col0)
