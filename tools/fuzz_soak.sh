#!/bin/bash
# fuzz_soak.sh FIRST LAST [search|moves]: the randomised differential tests under the seeds FIRST..LAST (ACX_FUZZ_SEED), one pytest run per
# seed; logs in gpurun_out/soak_SEED.log, one summary line per seed.  search (default): tests/test_gpu_search_fuzz.py (~8-14 s per seed on an
# MI355X); moves: the random-state tests of the move kernels and the env kernels against the oracle.
what=${3:-search}
for s in $(seq $1 $2); do
  if [ $what = moves ]; then
    ACX_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_gpu_moves.py tests/test_gpu_env.py -x -q -m gpu -k "random or wide_rows or config2_against_oracle or all_word_widths" > gpurun_out/soak_$s.log 2>&1
  else
    ACX_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_gpu_search_fuzz.py -x -q -m gpu > gpurun_out/soak_$s.log 2>&1
  fi
  echo "seed $s: $(grep -E 'passed|failed|error' gpurun_out/soak_$s.log | tail -1)"
done
