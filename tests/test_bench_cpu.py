"""bench.py's CPU-side legs (the oracle timed as `cpu_baseline`) run without a GPU; the GPU legs are covered by the driver's
own bench run.  The oracle is used here as what it is: the checker / baseline, never the product path."""
import numpy as np


def _states(n=8192, L=25):
    st = np.zeros((n, 2 * L), np.int8)
    st[:, :3] = [1, 2, 1]
    st[:, L:L + 3] = [2, 1, 2]
    return st


def test_cpu_baseline_legs_report_the_contract_fields():
    import bench

    one = bench.cpu_baseline(_states(), 0, budget_s=0.3)
    assert one["kind"] == "port" and one["cores"] == 1 and one["unit"] == "env-steps/s" and one["value"] > 0 and "sample" in one
    allc = one["all_cores"]
    assert allc["cores"] >= 1 and allc["value"] > 0 and allc["unit"] == "env-steps/s"


def test_algorithmic_bytes_per_step_is_4L_plus_7():
    import bench

    assert bench.ALGO_BYTES_PER_STEP == 4 * bench.L + 7 == 107  # state in + out, action, f32 reward, done, truncated


def test_search_roofline_prices_children_with_surveys_formula():
    import bench

    r = bench.search_roofline({"children": 332140812, "nodes": 100000001, "seconds": 0.0111}, "k", "none")
    f = 100000001 / 332140812
    assert abs(r["bytes_per_child"] - (64 + 72 * f)) < 1e-9 and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert abs(r["achieved"] - (64 + 72 * f) * 332140812 / 0.0111 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12


def test_python_numpy_leg_reports_steps_and_search_rates():
    import bench

    st = bench.ms_pool_at_L(25) if False else _states(64)  # the MS generator needs the GPU; any valid rows will do here
    leg = bench.cpu_python_baseline(st, budget_s=0.2)
    assert leg["value"] > 0 and leg["cores"] == 1 and leg["bfs_nodes_per_s"] > 0 and leg["greedy_search_nodes_per_s"] > 0


def test_timed_window_is_at_least_16384_steps_for_any_k():
    import bench

    for k in (1, 20, 1000, 16384, 50000):
        r = max(1, -(-bench.MIN_TIMED_STEPS // k))
        assert r * k >= bench.MIN_TIMED_STEPS and (r == 1 or (r - 1) * k < bench.MIN_TIMED_STEPS)
        p = max(1, min(r, bench.GRAPH_NODES_MAX // k))  # passes captured into one graph; the graph is launched ceil(r / p) times
        g = -(-r // p)
        assert p * k <= max(k, bench.GRAPH_NODES_MAX) and p * g >= r and p * (g - 1) < r


# ---- `python bench.py --gpus N` without a launcher: bench.py starts the N rank processes itself ---------------------------------
_STUB = """
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[1]
if mode == "ok":
    print("rank %d chatter" % rank)
    if rank == 0:
        print(json.dumps({"metric": "m", "n_gpus": world, "argv": sys.argv[2:]}))
elif mode == "rank1_dies":
    if rank == 1:
        sys.exit(5)
    time.sleep(60)   # the others sit in a collective that will never complete
elif mode == "mislabelled":
    if rank == 0:
        print(json.dumps({"metric": "m", "n_gpus": 1}))
"""


def _stub(tmp_path):
    import sys

    f = tmp_path / "stub.py"
    f.write_text(_STUB)
    return [sys.executable, str(f)]


def test_launcher_relays_rank0s_one_json_line(tmp_path, capfd):
    import json

    import bench

    rc = bench.launch_ranks(3, ["--steps", "7"], child=_stub(tmp_path) + ["ok"])
    out = capfd.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert rc == 0 and len(lines) == 1
    got = json.loads(lines[0])
    assert got["n_gpus"] == 3 and got["argv"] == ["--steps", "7"]
    assert "rank 0 chatter" not in out  # only the JSON line reaches stdout


def test_launcher_fails_when_any_rank_fails_and_ends_the_others(tmp_path, capfd):
    import time

    import bench

    t0 = time.time()
    rc = bench.launch_ranks(3, [], child=_stub(tmp_path) + ["rank1_dies"], grace_s=1.0)
    io = capfd.readouterr()
    assert rc == 5 and "{" not in io.out and "rank 1 exited with code 5" in io.err
    assert time.time() - t0 < 30  # ranks 0 and 2 were terminated, not waited for


def test_launcher_refuses_a_mislabelled_line(tmp_path, capfd):
    import bench

    assert bench.launch_ranks(2, [], child=_stub(tmp_path) + ["mislabelled"]) == 1
    io = capfd.readouterr()
    assert "{" not in io.out and "n_gpus = 1" in io.err


def test_bench_gpus_2_on_a_box_without_two_gpus_fails_loudly():
    """the whole path through the real file: `python bench.py --gpus 2` must start two ranks, and where two GPUs are not there it must
    exit non-zero WITHOUT printing a result line (round 3: it printed `n_gpus: 1`)"""
    import os
    import subprocess
    import sys

    import bench

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-search", "--no-extras", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "n_gpus" not in r.stdout, (r.returncode, r.stdout[-300:])
    assert "device(s)" in r.stderr and "cannot be measured here" in r.stderr, r.stderr[-600:]


def test_bench_refuses_a_gpus_flag_that_contradicts_the_launcher():
    import os
    import subprocess
    import sys

    import bench

    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, bench.__file__, "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""
