# ab_sweep.sh NAME...: tools/ms_sweep_warm.py bfs 1e6 for libacx and each variant, interleaved twice
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in libacx "$@"; do
    lib=$GRAFT_REPO_ROOT/ac-solver_amd/lib/$v.so
    echo "== $v (rep $rep)" >> gpurun_out/ab/sweep.log
    ACX_LIB=$lib timeout 300 python tools/ms_sweep_warm.py bfs 1e6 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/ab/sweep.log
  done
done
cat gpurun_out/ab/sweep.log
