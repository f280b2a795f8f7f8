mkdir -p gpurun_out/r4_final
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r4_final/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_final/tests.log
s=$(date +%s); python bench.py > gpurun_out/r4_final/bench.json 2> gpurun_out/r4_final/bench.err; echo "bench rc=$? seconds=$(( $(date +%s) - s ))" >> gpurun_out/r4_final/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_final/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r4_final/tests.log
tail -5 gpurun_out/r4_final/tests.log
