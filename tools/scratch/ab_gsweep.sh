mkdir -p gpurun_out/abg
for rep in 1 2; do
  for v in libacx "$@"; do
    echo "== $v (rep $rep)" >> gpurun_out/abg/sweep.log
    ACX_LIB=$GRAFT_REPO_ROOT/ac-solver_amd/lib/$v.so timeout 300 python tools/ms_sweep_warm.py greedy 1e6 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/abg/sweep.log
  done
done
cat gpurun_out/abg/sweep.log
timeout 900 python -m pytest tests/test_gpu_sweeps.py tests/test_gpu_search.py tests/test_gpu_search_fuzz.py -m gpu -x -q -k "greedy or sweep or fuzz or many or miller or fixtures or pool" > gpurun_out/abg/tests.log 2>&1; tail -2 gpurun_out/abg/tests.log
