#!/bin/bash
# Round-5 evidence for the env-step roofline AT THE METRIC'S OWN SIZE (65 536 envs): everything comes from the
# bench command itself -- the same hipGraph replay, the same 128-row rollout ring --, not from a separate driver:
#   * the plain line (HIP events around the timed region),
#   * rocprofv3 --kernel-trace --stats of the same command: begin-to-end of every dispatch of the replayed graph,
#   * --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own over tools/env_roofline.py at the same size (gfx950: 2 x FETCH_SIZE + WRITE_SIZE),
#   * in-kernel wave stamps of the -DACX_STEP_STAMP build of this round's kernel (tools/step_stamps.py, same graph replay).
# On the GPU box:  bash tools/profile_env_r5.sh   -> gpurun_out/r5env/summary.json (-> profiles/r5_env_step_roofline.json)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5env
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-search --no-extras --no-cpu-baseline"
python3 $R/bench.py $ARGS > $O/plain.json 2> $O/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $ARGS > $O/kt.json 2> $O/kt.err
# (the counter passes: rocprofv3 --pmc crashes inside the launch when bench.py enqueues its 17 340 launches without a pause -- replayed
# graph or eager alike, on this image --, so they run tools/env_roofline.py: the same kernel, batch and 128-row rollout ring, a
# synchronisation every 20-400 launches; the bytes a launch moves do not depend on how it was launched)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/env_roofline.py 65536 400 int8 128 1 > $O/fetch.json 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/env_roofline.py 65536 400 int8 128 1 > $O/write.json 2> $O/write.err
cd $R
[ -f ac-solver_amd/lib/var_stamp.so -a ac-solver_amd/lib/var_stamp.so -nt ac-solver_amd/csrc/acx_step.hip ] || bash tools/build_variant.sh stamp -DACX_STEP_STAMP > $O/build_stamp.log 2>&1
ACX_LIB=$R/ac-solver_amd/lib/var_stamp.so python3 tools/step_stamps.py 65536 128 10 > $O/stamps_65536.json 2> $O/stamps.err
python3 tools/summarize_env_r5.py $O > $O/summary.json 2> $O/summary.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/r5_env_step_65536_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
cat $O/summary.json
