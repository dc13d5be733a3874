"""bench.py's own launcher on the CPU: `python bench.py --gpus N` without torchrun must start N fresh rank processes (before
the parent makes any GPU call -- it never does), the ranks must meet over gloo on 127.0.0.1 and shard the columns by rank;
with a launcher's environment the flag must agree with WORLD_SIZE.  --dry-run stops before the first GPU call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return e


def test_gpus_flag_starts_that_many_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--ncol", "1000"], capture_output=True, text=True,
                         timeout=300, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2
    assert d["shards_rank_device_col0_ncol"] == [[0, 0, 0, 1000], [1, 1, 1000, 1000]]


def test_device_map_for_rehearsals_on_one_gpu():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dry-run", "--ncol", "64", "--device-map", "0,0,0"],
                         capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and [s[1] for s in d["shards_rank_device_col0_ncol"]] == [0, 0, 0]
    assert [s[2] for s in d["shards_rank_device_col0_ncol"]] == [0, 64, 128]


def test_launcher_world_size_must_match_the_flag():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_single_rank_dry_run():
    out = subprocess.run([sys.executable, BENCH, "--dry-run"], capture_output=True, text=True, timeout=120, env=_env())
    assert out.returncode == 0
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1
