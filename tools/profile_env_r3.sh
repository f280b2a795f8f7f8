#!/bin/bash
# Round-3 evidence for the env-step roofline.  On the GPU box:  bash tools/profile_env_r3.sh
# For N = 65 536, 2^20 and 4 Mi envs: the unprofiled HIP-event time per launch, rocprofv3 --kernel-trace --stats of the same
# command, and --pmc FETCH_SIZE / --pmc WRITE_SIZE in passes of their own; then the in-kernel stamps of the diagnostic build.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3env
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N in 65536 1048576 4194304; do
  K=40; [ $N = 65536 ] && K=400
  python3 $R/tools/env_roofline.py $N $K > $O/plain_$N.json 2> $O/plain_$N.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$N -- python3 $R/tools/env_roofline.py $N $K > $O/kt_$N.json 2> $O/kt_$N.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$N -- python3 $R/tools/env_roofline.py $N $K > $O/fetch_$N.json 2> $O/fetch_$N.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$N -- python3 $R/tools/env_roofline.py $N $K > $O/write_$N.json 2> $O/write_$N.err
done
python3 $R/tools/env_roofline.py 1048576 40 float32 > $O/plain_f32_1048576.json 2>> $O/plain_1048576.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_f32_1048576 -- python3 $R/tools/env_roofline.py 1048576 40 float32 > $O/kt_f32_1048576.json 2> $O/kt_f32.err
cd $R
[ -f ac-solver_amd/lib/var_stamp.so ] || bash tools/build_variant.sh stamp -DACX_STEP_STAMP > $O/build_stamp.log 2>&1
ACX_LIB=$R/ac-solver_amd/lib/var_stamp.so python3 tools/step_stamps.py 65536 100 10 > $O/stamps_65536.json 2> $O/stamps.err
ACX_LIB=$R/ac-solver_amd/lib/var_stamp.so python3 tools/step_stamps.py 1048576 20 5 > $O/stamps_1048576.json 2>> $O/stamps.err
python3 tools/summarize_env_r3.py $O > $O/summary.json 2> $O/summary.err
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*counter_collection.csv" -size +5M -delete
cat $O/summary.json
rocprofv3 -L 2>/dev/null | grep -i -E "dram|mall|hbm|EA0_RD|EA0_WR" | head -40 > $O/counters_list.txt
