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
