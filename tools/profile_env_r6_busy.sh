#!/bin/bash
# A profiler-side BUSY time per k_env_step launch at the metric's size: rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
# over tools/env_roofline.py at 65 536 envs and at 4 Mi envs.  SQ_BUSY_CYCLES only counts while waves are on the chip (GRBM_GUI_ACTIVE
# also spans the profiler's per-dispatch work: it reads 12.6 us for a launch that the bench replays every 3.4 us); its unit (which SQ
# instances are summed) is calibrated on the 4 Mi-env launch, whose duration the kernel trace and the HIP events agree on.
#   bash tools/profile_env_r6_busy.sh   -> gpurun_out/r6busy/busy.json (merged into profiles/r6_env_step_roofline.json by tools/summarize_env_r6_busy.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6busy
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/busy_65536 -- python3 $R/tools/env_roofline.py 65536 400 int8 128 1 > $O/busy_65536.json 2> $O/busy_65536.err
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/busy_4194304 -- python3 $R/tools/env_roofline.py 4194304 40 int8 8 1 > $O/busy_4194304.json 2> $O/busy_4194304.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/waves_65536 -- python3 $R/tools/env_roofline.py 65536 400 int8 128 1 > $O/waves_65536.json 2> $O/waves_65536.err
cd $R
python3 tools/summarize_env_r6_busy.py $O > $O/busy.json 2> $O/busy.err
find $O -name "*counter_collection.csv" -delete
find $O -name "*.db" -delete
cat $O/busy.json
