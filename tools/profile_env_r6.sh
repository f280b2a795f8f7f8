#!/bin/bash
# Round-6 evidence for the env-step roofline, all on this round's acx_step.hip:
#   65 536 envs (the metric's size): the bench command plain and under rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE, WRITE_SIZE and
#     GRBM_GUI_ACTIVE in passes of their own over tools/env_roofline.py (same kernel, batch, 128-row ring) -- GRBM_GUI_ACTIVE / 8 XCDs /
#     clock = a profiler-side busy time per launch to set beside the in-kernel stamps; wave stamps of the -DACX_STEP_STAMP build;
#   1 Mi and 4 Mi envs (int8 rows; 4 Mi is the HBM-honest size) and 131 072 envs with f32 observation rows (BASELINE config 5's per-GPU
#     shape): HIP events, kernel trace and the two traffic passes of tools/env_roofline.py.
# On the GPU box:  bash tools/profile_env_r6.sh   -> gpurun_out/r6env/summary.json (-> profiles/r6_env_step_roofline.json)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6env
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-search --no-extras --no-cpu-baseline"
python3 $R/bench.py $ARGS > $O/plain.json 2> $O/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $ARGS > $O/kt.json 2> $O/kt.err
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_65536 -- python3 $R/tools/env_roofline.py 65536 400 int8 128 1 > $O/pmc_${C}_65536.json 2> $O/pmc_${C}_65536.err
done
for CFG in "1048576 int8" "4194304 int8" "131072 float32"; do
  set -- $CFG
  N=$1; DT=$2
  python3 $R/tools/env_roofline.py $N 40 $DT > $O/plain_${N}_$DT.json 2> $O/plain_${N}_$DT.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${N}_$DT -- python3 $R/tools/env_roofline.py $N 40 $DT > $O/kt_${N}_$DT.json 2> $O/kt_${N}_$DT.err
  for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_${C}_${N}_$DT -- python3 $R/tools/env_roofline.py $N 40 $DT 8 1 > $O/pmc_${C}_${N}_$DT.json 2> $O/pmc_${C}_${N}_$DT.err
  done
done
cd $R
[ -f ac-solver_amd/lib/var_stamp.so -a ac-solver_amd/lib/var_stamp.so -nt ac-solver_amd/csrc/acx_step.hip ] || bash tools/build_variant.sh stamp -DACX_STEP_STAMP > $O/build_stamp.log 2>&1
ACX_LIB=$R/ac-solver_amd/lib/var_stamp.so python3 tools/step_stamps.py 65536 128 10 > $O/stamps_65536.json 2> $O/stamps.err
python3 tools/summarize_env_r6.py $O > $O/summary.json 2> $O/summary.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r6_env_step_65536_kernel_stats.csv
for CFG in 1048576_int8 4194304_int8 131072_float32; do
  cp $(find $O/kt_$CFG -name "*kernel_stats.csv" | head -1) $O/r6_env_step_${CFG}_kernel_stats.csv
done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*.db" -delete
cat $O/summary.json
