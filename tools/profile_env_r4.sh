#!/bin/bash
# Round-4 evidence for the env-step roofline (BENCH_r03 said 67.3 us per 4 Mi-env launch, profiles/r3_* 76.0 us).
# tools/env_roofline.py now warms the device up for 0.3 s and rotates its outputs over 8 rollout rows, as bench.py does.  Per
# batch size: the unprofiled HIP-event time (median of 5 blocks of 40 launches), rocprofv3 --kernel-trace --stats of the same
# command (the trace CSV is reduced to the steady launches by tools/summarize_env_r4.py) and --pmc FETCH_SIZE / WRITE_SIZE passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4env
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N in 1048576 4194304; do
  python3 $R/tools/env_roofline.py $N 40 > $O/plain_$N.json 2> $O/plain_$N.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$N -- python3 $R/tools/env_roofline.py $N 40 > $O/kt_$N.json 2> $O/kt_$N.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$N -- python3 $R/tools/env_roofline.py $N 40 int8 8 1 > $O/fetch_$N.json 2> $O/fetch_$N.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$N -- python3 $R/tools/env_roofline.py $N 40 int8 8 1 > $O/write_$N.json 2> $O/write_$N.err
done
python3 $R/tools/summarize_env_r4.py $O > $O/summary.json 2> $O/summary.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
cat $O/summary.json
