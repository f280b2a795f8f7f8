mkdir -p gpurun_out/gchain
L=gpurun_out/gchain/ab2.log
for hm in 256 384 512 640 1024; do echo "== chained hand_min $hm" >> $L; ACX_GREEDY_HAND_MIN=$hm timeout 300 python3 tools/greedy_only.py 1e7 3 2>&1 | tail -2 >> $L; done
for hm in 256 512 1024; do echo "== chained hand_min $hm budget 1e6" >> $L; ACX_GREEDY_HAND_MIN=$hm timeout 300 python3 tools/greedy_only.py 1e6 4 2>&1 | tail -2 >> $L; done
cat $L
