R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/shardw
cd /tmp && export TMPDIR=/tmp
for w in 1 2 4; do
AMD_SERIALIZE_KERNEL=3 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/shardw/w$w -o s -- python3 $R/tools/shard_threads_only.py $w 1e8 3 > $R/gpurun_out/shardw/w$w.log 2>&1
echo "== world $w"; grep "world" $R/gpurun_out/shardw/w$w.log | tail -3
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/shardw/w$w/s_kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"]
    if "k_shard" in n or "fill" in n or "copy" in n.lower():
        t=float(r["TotalDurationNs"])/3e6; tot+=t
        print(f"   {n[:60]:60s} {int(r['Calls'])//3:6d} calls/search {t:8.2f} ms/search")
print(f"   total {tot:.2f} ms per search (all ranks)")
PY
done
find $R/gpurun_out/shardw -name "*kernel_trace.csv" -delete
