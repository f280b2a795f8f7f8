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
