for w in 7 4 3 2 7; do echo "== greedy sweep workers $w"; ACX_SWEEP_WORKERS=$w python3 tools/ms_sweep_warm.py greedy 1e6 2>&1 | tail -2; done
