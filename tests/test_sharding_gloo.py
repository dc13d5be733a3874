"""N > 1 path on CPU: two processes over gloo, each stepping its shard of a perturbed SHEBA ensemble; the
concatenation of the shards must equal the single-process ensemble bit for bit (columns are independent, so sharding
must not change any result) and the max-over-ranks / sum reductions of bench.py must work.  The CPU oracle stands in for
the device here -- the partition and reduction logic is what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from samsim_amd import testcases as tcs
from samsim_amd.shard import shard_range
from tests.helpers import load_checkpoint, sheba_forcing
from tests.oracle_lib import oracle_solver

NCOL, NSTEPS = 6, 300


def run_shard(col0, n):
    cfg, _ = tcs.testcase4(1)
    st1, clock = load_checkpoint("tc4_spunup_state.npz")
    o = oracle_solver(cfg, n)
    dT, ps = tcs.ensemble_perturbation(n, col0)
    o.set_forcing(*sheba_forcing(), dT, ps)
    o.set_state(st1.replicate(n))
    o.set_clock(**clock)
    o.step(NSTEPS)
    return o.get_state()


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    col0, n = shard_range(NCOL, rank, world)
    st = run_shard(col0, n)
    np.savez(os.path.join(tmp, f"shard{rank}.npz"), lay=st.lay, scal=st.scal, n_active=st.n_active, col0=col0)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == world
    s = torch.tensor([float(n)], dtype=torch.float64)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    assert float(s[0]) == NCOL
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_the_ensemble():
    for total, world in [(1 << 20, 8), (10, 3), (7, 8)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == total
        for (a, n), (b, _) in zip(spans, spans[1:]):
            assert a + n == b


def test_two_rank_gloo_shards_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole = run_shard(0, NCOL)
    for r in range(2):
        z = np.load(tmp_path / f"shard{r}.npz")
        c0, n = int(z["col0"]), z["lay"].shape[2]
        assert np.array_equal(z["lay"], whole.lay[:, :, c0:c0 + n])
        assert np.array_equal(z["scal"], whole.scal[:, c0:c0 + n])
        assert np.array_equal(z["n_active"], whole.n_active[c0:c0 + n])
    # the perturbation reaches the columns (snow mass responds to the scaled precipitation)
    assert len(np.unique(whole.sc("T2m"))) == NCOL and len(np.unique(whole.sc("m_snow"))) > 1
