"""bench.py on the GPU box: the launcher path (`--gpus N` without a launcher) and the N > 1 code path of the secondary
measurements, driven on the one GPU a box has (RCCL at world 1 through ACX_BENCH_FORCE_DIST)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_fails_loudly_instead_of_reporting_one_gpu():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the two-rank run is legitimate here")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-search", "--no-extras", "--no-cpu-baseline"],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "n_gpus" not in r.stdout, (r.returncode, r.stdout[-300:])
    assert "cannot be measured here" in r.stderr, r.stderr[-800:]


@pytest.mark.gpu
def test_the_multi_rank_code_path_runs_on_one_gpu_and_reports_its_diagnostics():
    """ACX_BENCH_FORCE_DIST: process group (RCCL, world 1), the sharded search through TorchDistComm on the shared communicator and on
    a communicator of its own for the masks, per-stage device times of a chunk, cpu_baseline on the line although the run is
    'distributed'."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "20", "--warmup", "5", "--no-extras", "--search-budget", "3000000"],
                       env=_env(ACX_BENCH_FORCE_DIST="1", ACX_BENCH_STRONG_BUDGET="30000000", ACX_BENCH_CPU_SECONDS="1", MASTER_PORT="29533"),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["cpu_baseline"]["value"] > 0 and out["roofline"]["frac"] > 0
    s = out["search"]
    assert "error" not in s, s.get("error")
    one = s["bfs_sharded"]
    assert 3000000 <= one["nodes"] < 3000012 and one["rccl_ranks_seen"] == 1 and one["backend"] == "nccl"
    tl = one["timeline"]
    for k in ("expand_us", "all_to_all_us", "insert_us", "mask_all_reduce_us", "commit_us"):
        assert tl[k] is not None and tl[k] >= 0, (k, tl)
    tl4 = s["bfs_sharded_strong"]["timeline"]  # (enough chunks per level for a period)
    assert tl4["chunk_period_us"] > 0 and tl4["overlap_effective"] > 0, tl4
    assert set(one["by_mask_group"]) == {"shared", "own"} and one["mask_all_reduce_group"] in ("shared", "own")
    assert one["collectives"]["all_to_all_calls"] > 0 and one["collectives"]["mask_all_reduce_calls"] > 0
    assert 30000000 <= s["bfs_sharded_strong"]["nodes"] < 30000012 and s["bfs_sharded_strong"]["budget"] == 30000000
    assert s["bfs_ms_sweep"]["solved"] == 278 and s["greedy_ms_sweep"]["solved"] == 533
