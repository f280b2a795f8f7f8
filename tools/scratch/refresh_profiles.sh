set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "gpu tests rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --steps 20 --warmup 5 > $O/bench_line_steps20.json 2> $O/bench_steps20.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-search > $O/bench_line_under_rocprof.json 2> $O/bench_kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bfs_kt -- python3 $R/tools/bfs_only.py 1e8 > $O/bfs_kt.log 2>&1
bash $R/tools/pmc_passes.sh $O/pmc python3 $R/tools/bfs_only.py 1e8
python3 $R/tools/pmc_sum.py $O/pmc > $O/pmc_summary.txt 2>&1
find $O -name "*.csv" -size +20M -delete
ls -la $O
