"""Timeline of the k_shard_* kernels of every search in a rocprofv3 --kernel-trace results .db: per search the span, the sum of
kernel time, and for the full-size chunks the mean duration of each kernel and the mean start-to-start period of k_shard_insert."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = list(c.execute("select name,start,end,queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_shard_seed" in r[0]] + [len(rows)]
verbose = len(sys.argv) > 2
for si in range(len(idx) - 1):
    seg = [r for r in rows[idx[si]:idx[si + 1]] if "k_shard" in r[0]]
    t0 = seg[0][1]
    short = lambda n: n.split("k_shard_")[1].split("<")[0].split("(")[0]
    ins = [r for r in seg if short(r[0]) == "insert" and r[2] - r[1] > 400000]
    per = [(b[1] - a[1]) / 1e3 for a, b in zip(ins, ins[1:]) if b[1] - a[1] < 1500000]
    dur = collections.defaultdict(list)
    for r in seg:
        if r[2] - r[1] > 30000: dur[short(r[0])].append((r[2] - r[1]) / 1e3)
    big = {k: sum(sorted(v)[len(v) // 2:]) / len(sorted(v)[len(v) // 2:]) for k, v in dur.items()}
    print(f"search {si}: span {(seg[-1][2] - t0) / 1e6:.2f} ms, kernel sum {sum(r[2] - r[1] for r in seg) / 1e6:.2f} ms, queues {sorted(set(r[3] for r in seg))}, "
          f"insert period {sum(per) / max(1, len(per)):.0f} us over {len(per)}; upper-half mean us: " + " ".join(f"{k} {v:.0f}" for k, v in sorted(big.items())))
    if verbose and si == int(sys.argv[2]):
        for r in seg:
            if r[2] - r[1] > 20000: print(f"   {(r[1] - t0) / 1e6:8.3f} {(r[2] - t0) / 1e6:8.3f} {(r[2] - r[1]) / 1e3:8.1f} q{r[3]} {short(r[0])}")
