#!/bin/bash
# fuzz soak: the differential fuzz tests under other seeds
for s in $(seq $1 $2); do
  ACX_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_gpu_search_fuzz.py -x -q -m gpu > gpurun_out/soak_$s.log 2>&1
  echo "seed $s: $(grep -E 'passed|failed|error' gpurun_out/soak_$s.log | tail -1)"
done
