mkdir -p gpurun_out/r4_13
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r4_13/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_13/tests.log
python bench.py > gpurun_out/r4_13/bench.json 2> gpurun_out/r4_13/bench.err; echo "bench rc=$?" >> gpurun_out/r4_13/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_13/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r4_13/tests.log
bash tools/profile_r4.sh greedy bfs > gpurun_out/r4_13/prof.log 2>&1
tail -6 gpurun_out/r4_13/tests.log
