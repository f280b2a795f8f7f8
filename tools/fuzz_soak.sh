#!/bin/bash
# fuzz_soak.sh FIRST LAST: the differential fuzz tests (tests/test_gpu_search_fuzz.py) under the seeds FIRST..LAST (ACX_FUZZ_SEED), one
# pytest run per seed (~5 s each on an MI355X); logs in gpurun_out/soak_SEED.log, one summary line per seed.
for s in $(seq $1 $2); do
  ACX_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_gpu_search_fuzz.py -x -q -m gpu > gpurun_out/soak_$s.log 2>&1
  echo "seed $s: $(grep -E 'passed|failed|error' gpurun_out/soak_$s.log | tail -1)"
done
