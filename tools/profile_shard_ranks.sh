#!/bin/bash
# Device work of ALL ranks of one sharded bfs at world W, measured on ONE GPU (W thread ranks share the device, kernels serialised so
# that a kernel's duration is its exclusive time): gpurun_out/shard_ranks/<tag>_shard_thread_ranks_device_work.txt is the table of
#   bash tools/profile_shard_ranks.sh [tag] [budget] [searches] ["1 2 4 8"] [log2 of the global parents per chunk]   (on the GPU box; ~1 min)
# Not a scaling measurement: it says what a rank of a real W-GPU run has to do = total / W (+ the exchange over xGMI instead of the
# ThreadComm copies).
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r5}; budget=${2:-1e8}; reps=${3:-3}; worlds=${4:-"1 2 4 8"}; export CHUNK_LOG2=${5:-21}
export TMPDIR=/tmp
out=gpurun_out/shard_ranks; rm -rf $out; mkdir -p $out
for w in $worlds; do
    AMD_SERIALIZE_KERNEL=3 rocprofv3 --kernel-trace --stats --output-format csv -d $out/w$w -o t -- python3 tools/shard_threads_only.py $w $budget $reps $CHUNK_LOG2 > $out/w$w.log 2>&1
    grep "^world" $out/w$w.log | cut -c1-400
done
# (the table goes to gpurun_out/, which travels back from the GPU box; copy it to profiles/<tag>_shard_thread_ranks_device_work.txt)
python3 tools/shard_ranks_table.py $out $reps $worlds > $out/${tag}_shard_thread_ranks_device_work.txt
cat $out/${tag}_shard_thread_ranks_device_work.txt
