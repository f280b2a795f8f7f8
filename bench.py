#!/usr/bin/env python3
"""bench.py -- headline benchmark of the AC hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 65 536 batched ACEnv.step per GPU, Miller-Schupp initial states
(pool of 1190 presentations, env e starts from pool[e mod 1190]) re-embedded at max_relator_length = 25,
uniform random actions from a pre-generated device tape, horizon 1000, gymnasium-style autoreset.
A "step" is ONE launch of the env kernel over the whole batch: it reads the packed state + the action
and writes the new state, the int8 observation [N, 50], the f32 reward and the done / truncated flags
into row t of PPO-style rollout buffers that are already resident in HBM.

One process per GPU (torch.distributed / RCCL only for the barrier and the MAX over ranks: the
environments are independent, so there is no data-path collective -> "scaling": "weak").
Rank 0 prints ONE JSON line.  `roofline` prices the env kernel against HBM with the ALGORITHMIC bytes
of one step (4L + 7 = 107 B at L = 25); `search.bfs.roofline` / `search.greedy_search.roofline`
price the frontier kernels with SURVEY 8(d)'s (64 + 72 f) B per generated child; `cpu_baseline` times the
CPU oracle (plain C port of the reference's algorithm, oracle/ac_oracle.c) on one host core and on all of
them, and the pure Python / NumPy restatement (oracle/ac_numpy.py), on bounded samples of the same workloads.

Timed region: R = ceil(16384 / K) passes over the K steps (a window of >= 50 ms whatever K is), back to back between
barrier + synchronize on both sides; `ms_per_step` = window / (R * K), `repeats` = R.  The passes are captured into
hipGraphs of up to 1024 kernel nodes (`graph_nodes`; K = 20: 51 passes per graph, launched 17 times): two graph launches
leave a bubble of ~6 us between them on the GPU, which a 20-node graph would pay every 20 steps.  A 20-step and a 1000-step
run therefore report the per-step time of the kernel chain, not one graph launch's fixed cost spread over 20 steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "ac-solver_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

L = 25
N_ENVS = 65536
HORIZON = 1000
ALGO_BYTES_PER_STEP = 4 * L + 7  # state in + out (2 x 2L), action 1, reward f32 4, done 1, truncated 1
SEARCH_TIMEOUT_S = int(os.environ.get("ACX_BENCH_SEARCH_TIMEOUT", "240"))  # multi-rank runs: how long the RCCL-backed secondary measurements may take
HBM_PEAK_GBS = 8000.0            # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
MIN_TIMED_STEPS = 16384          # the K steps are repeated ceil(MIN_TIMED_STEPS / K) times: a window of >= 50 ms
MIN_ROLLOUT_ROWS = 128           # rows of the rollout buffers the timed steps rotate over, whatever --steps is: 128 x 3.7 MB = 470 MB at 65 536
                                 # envs, beyond the 256 MiB Infinity Cache -- the observation / reward / flag streams of a launch HAVE to reach HBM
                                 # (a PPO rollout buffer of horizon 1000 is larger still); --steps 20 alone kept the whole working set cache resident
GRAPH_NODES_MAX = 1024           # passes of the same K steps captured into ONE hipGraph (K = 20: 51 passes): consecutive graph
                                 # launches leave a ~6 us bubble on the GPU, which a 20-node graph pays every 20 steps


def search_roofline(stats, kernel, traffic_key):
    """one generated child costs 64 B (one random table sector) + f * 72 B when it is new (slot, frontier append,
    parent / action / length record), f = nodes / children.  achieved = those bytes / the device time of the search."""
    children, nodes, dev_s = stats["children"], stats["nodes"], stats["seconds"]
    f = nodes / max(children, 1)
    algo = (64.0 + 72.0 * f) * children
    out = {"bound": "hbm", "achieved": algo / dev_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / dev_s / 1e9 / HBM_PEAK_GBS,
           "traffic": None, "kernel": kernel, "algorithmic_bytes": algo, "bytes_per_child": 64.0 + 72.0 * f, "children": children,
           "new_fraction": f, "device_seconds": dev_s}
    tf = os.path.join(ROOT, "profiles", "search_traffic.json")
    if os.path.exists(tf):  # HBM bytes of the same search from rocprofv3 --pmc passes (profiles/README.md): provenance stated
        with open(tf) as fh:
            t = json.load(fh).get(traffic_key)
        if t and t.get("children") == children:
            out["traffic"] = t["hbm_bytes"]
            out["traffic_source"] = t["source"]
    return out


def ms_pool_at_L(L):
    """The 1190 Miller-Schupp presentations (n = 1..7, max_w_len = 7, generator order) at width L."""
    from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

    pool = []
    for n in range(1, 8):
        d = generate_miller_schupp_presentations(n, 7)
        for w in range(1, 8):
            for p in d[w]:
                p = np.asarray(p, np.int8)
                half = len(p) // 2
                row = np.zeros(2 * L, np.int8)
                for h in (0, 1):
                    word = p[h * half:(h + 1) * half]
                    word = word[word != 0]
                    row[h * L:h * L + len(word)] = word
                pool.append(row)
    return np.stack(pool)


CPU_SCALE = float(os.environ.get("ACX_BENCH_CPU_SECONDS", "0")) / 12.0 or 1.0  # (tests: shrink the CPU legs)


def cpu_baseline(states, seed, budget_s=12.0):
    """Oracle ACEnv.step (1 thread) on a bounded sample of the same workload: 4096-env slices x 64 steps."""
    from oracle import ac_oracle as O

    rng = np.random.default_rng(seed)
    n = 4096
    st = np.ascontiguousarray(states[:n]).copy()
    cnt = np.zeros(n, np.int32)
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        tape = rng.integers(0, 12, size=(64, n), dtype=np.uint8)
        O.env_rollout(st, cnt, HORIZON, tape, want_outputs=True)
        done = cnt >= HORIZON
        if done.any():  # keep the sample in the same regime as the GPU run (autoreset at the horizon)
            st[done] = states[:n][done]
            cnt[done] = 0
        steps += 64 * n
    dt = time.perf_counter() - t0
    out = {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{steps} ACEnv.step calls ({n} envs, MS initial states, L={L}) by oracle/ac_oracle.c in {dt:.1f} s"}
    out["all_cores"] = cpu_baseline_all_cores(states, seed + 1, budget_s=budget_s / 2)
    return out


def cpu_baseline_all_cores(states, seed, budget_s=6.0):
    """The same oracle loop on every host core this process may use (one thread per core; ctypes releases the GIL
    during the C call), each thread with its own slice of environments."""
    import threading

    from oracle import ac_oracle as O

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = 4096
    rng = np.random.default_rng(seed)
    work = []
    for t in range(cores):
        lo = (t * n) % max(len(states) - n, 1)
        bufs = (np.empty((64, n), np.int32), np.empty((64, n), np.uint8), np.empty((64, n), np.uint8))
        work.append([np.ascontiguousarray(states[lo:lo + n]).copy(), np.zeros(n, np.int32), rng.integers(0, 12, size=(64, n), dtype=np.uint8), 0, bufs,
                     np.ascontiguousarray(states[lo:lo + n])])
    deadline = time.perf_counter() + budget_s

    def run(w):
        while time.perf_counter() < deadline:
            O.env_rollout(w[0], w[1], HORIZON, w[2], want_outputs=True, out=w[4])
            done = w[1] >= HORIZON
            if done.any():  # same regime as the one-core sample: back to the initial state at the horizon
                w[0][done] = w[5][done]
                w[1][done] = 0
            w[3] += 64 * n

    threads = [threading.Thread(target=run, args=(w,)) for w in work]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    steps = sum(w[3] for w in work)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": cores,
            "sample": f"{steps} ACEnv.step calls, {cores} threads x {n} envs, in {dt:.1f} s"}


def ak3_at_L():
    """AK(3) = <x,y | x^3 = y^4, xyx = yxy> at max_relator_length = 25 (BASELINE config 3)."""
    p = np.zeros(2 * L, np.int8)
    p[:7] = [1, 1, 1, -2, -2, -2, -2]
    p[L:L + 6] = [1, 2, 1, -2, -1, -2]
    return p


STRONG_BUDGET = int(os.environ.get("ACX_BENCH_STRONG_BUDGET", 4 * 10**8))        # the 1 -> 8 GPU strong-scaling search: >= 5e7 nodes per rank at 8 ranks (the 1e8 search keeps 1.25e7: latency-bound by construction)


def search_numbers(world, rank, dev, budget, use_dist=False, out=None):
    """BFS nodes/s with the frontier sharded over `world` GPUs (one search, strong scaling) and, on one GPU, the
    fused single-GPU frontier (acx_search) for bfs and greedy_search.  AK(3) at L=25 does not trivialise, so the
    searches run to the node budget.  `out` is filled stage by stage (the watchdog of a multi-rank run prints what is there)."""
    import gc

    import torch
    import torch.distributed as dist

    from ac_solver import _acx
    from ac_solver.search._common import run_search
    from ac_solver.search.sharded import SingleComm, TorchDistComm, bfs_sharded

    out = {} if out is None else out
    p = ak3_at_L()
    if use_dist and world == 1:  # the forced one-rank run (ACX_BENCH_FORCE_DIST): route through the communicator as a multi-rank run does
        import ac_solver.search.sharded as _sh

        _sh._FORCE_EXCHANGE = True
    comms = {"shared": TorchDistComm(dev) if use_dist else SingleComm()}

    def timed_sharded(comm, b, reps=3):
        """median of `reps` timed searches (MAX over the ranks each) behind a full-size warm-up call; no garbage collection while the clock runs
        (as timeit does: a full collection of this process's heap takes 25-30 ms and landed in one search out of four)"""
        t0 = time.perf_counter()
        bfs_sharded(p, b, comm=comm)  # (chunks: bfs_sharded's default, 2^21 global parents, 2^23 from 8 ranks on) warm-up at full size: allocator (GBs of first-time hipMalloc), kernels, communicator
        torch.cuda.synchronize()
        first_call = time.perf_counter() - t0
        if use_dist:
            dist.barrier()
        samples, split, st = [], [], None
        for _ in range(reps):
            gc.collect()
            gc.disable()
            t0 = time.perf_counter()
            ok, path, st = bfs_sharded(p, b, comm=comm, want_stats=True)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            gc.enable()
            samples.append(float(dt[0]))
            split.append((round(st.get("setup_seconds", 0.0), 4), round(st.get("loop_seconds", 0.0), 4)))
        secs = sorted(samples)[len(samples) // 2]
        # SURVEY 8(d)'s formula for the frontier ((64 + 72 f) B per generated child); the children are 12 per expanded parent
        children = 12 * st["expanded"]
        f_new = st["nodes"] / max(children, 1)
        algo = (64.0 + 72.0 * f_new) * children
        return {"nodes_per_s": st["nodes"] / secs, "nodes": st["nodes"], "seconds": secs, "samples_seconds": samples, "samples_setup_loop_seconds": split,
                "first_call_seconds": first_call, "levels": st["levels"], "chunks": st["chunks"], "expanded": st["expanded"], "n_gpus": world, "budget": b,
                "input": "AK(3) at max_relator_len=25, cyclical=False", "scaling": "strong",
                "collectives": {k[5:]: v for k, v in st.items() if k.startswith("comm_")},
                "exchange_bytes_per_chunk_per_rank": (st.get("comm_all_to_all_bytes", 0) / max(st.get("comm_all_to_all_calls", 1), 1)),
                "local_nodes_of_rank0": st.get("local_nodes"), "owner": "class hashes + inner letters of both relators (csrc/acx_owner.h): ~3/4 of the children are owned by the rank that makes them",
                "region_fill_q8": st.get("region_fill_q8"), "region_overflow_reruns": st.get("region_overflow_reruns", 0),
                "roofline": {"bound": "hbm", "achieved": algo / secs / 1e9 / world, "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                             "frac": algo / secs / 1e9 / world / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "k_shard_expand<u64> (expansion + the claims of the children born on their owner) + k_shard_insert<u64> (received records) + k_shard_commit_born / k_shard_commit<u64> (whole search, wall time)",
                             "algorithmic_bytes": algo, "bytes_per_child": 64.0 + 72.0 * f_new, "children": children}}

    def timeline_of(comm, b):
        """one more search with HIP events around every stage of every chunk (not a timed sample: the events cost a little)"""
        try:
            _, _, st = bfs_sharded(p, b, comm=comm, want_stats=True, timeline=True)
            return st.get("timeline")
        except Exception as e:  # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"}

    exchange_text = ("per chunk: ONE equal-split all-to-all of child-record regions (headers carry counts and the success / error words) + "
                     "ONE all-reduce of a 12-bit child mask per parent (RCCL); expansion + all-to-all of chunk k+1 run on a side stream "
                     "beside the dedup + commit of chunk k") if world > 1 else "none (world 1: every child is owned by the one rank and claims its table slot from the expansion kernel; no record is written)"
    one = timed_sharded(comms["shared"], budget)
    one.update({"exchange": exchange_text, "rccl_ranks_seen": world if use_dist else 0, "backend": dist.get_backend() if use_dist else None,
                "mask_all_reduce_group": "shared" if use_dist else None})
    tf = os.path.join(ROOT, "profiles", "search_traffic.json")
    if world == 1 and os.path.exists(tf):  # HBM bytes of the same search from rocprofv3 --pmc passes (profiles/README.md)
        with open(tf) as fh:
            t = json.load(fh).get("bfs_sharded_ak3_1e8")
        if t and t.get("children") == one["roofline"]["children"]:
            one["roofline"]["traffic"] = t["hbm_bytes"]
            one["roofline"]["traffic_source"] = t["source"]
    out["bfs_sharded"] = one
    # where a chunk period goes: device time of the expansion, the all-to-all, the dedup, the mask all-reduce and the commit per
    # full-size chunk, and how much of their sum the side stream hides (overlap_effective > 1)
    one["timeline"] = timeline_of(comms["shared"], budget)
    # the same search as ONE C call per rank (acx_bfs_sharded, csrc/acx_shard_run.hip): the chunk loop in C++ on the process group's own RCCL
    # communicator (ProcessGroupNCCL._comm_ptr -> acx_comm_rccl); the numbers above are the Python orchestration of the same engine
    def timed_native(b, reps=3):
        from ac_solver.search.sharded import NativeComm, bfs_sharded_native

        nat = NativeComm.from_process_group(device=dev) if (use_dist and world > 1) else None
        bfs_sharded_native(p, b, comm=nat)
        torch.cuda.synchronize()
        samples, st = [], None
        for _ in range(reps):
            if use_dist:
                dist.barrier()
            t0 = time.perf_counter()
            ok, path, st = bfs_sharded_native(p, b, comm=nat, want_stats=True)
            torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            samples.append(float(dt[0]))
        secs = sorted(samples)[len(samples) // 2]
        return {"nodes_per_s": st["nodes"] / secs, "nodes": st["nodes"], "seconds": secs, "samples_seconds": samples, "levels": st["levels"], "chunks": st["chunks"],
                "replicated_levels": st["replicated_levels"], "expanded": st["expanded"], "n_gpus": world, "budget": b, "setup_seconds": st["setup_seconds"],
                "loop_seconds": st["loop_seconds"], "collectives": {k: st[k] for k in ("all_to_all_calls", "all_to_all_bytes", "all_reduce_calls", "all_reduce_bytes")},
                "entry": "acx_bfs_sharded: one C call per rank" + (", collectives on the process group's RCCL communicator (acx_comm_rccl)" if nat is not None else " (one rank: no collective)")}

    try:
        out["bfs_sharded_native"] = timed_native(budget)
        out["bfs_sharded_native_strong"] = timed_native(STRONG_BUDGET, reps=2)
    except Exception as e:  # noqa: BLE001
        out["bfs_sharded_native"] = dict(out.get("bfs_sharded_native", {}), error=f"{type(e).__name__}: {e}")
    # the strong-scaling point of the 1 -> 8 GPU curve: the SAME 4e8-node search at every N
    try:
        out["bfs_sharded_strong"] = timed_sharded(comms["shared"], STRONG_BUDGET, reps=3)
        out["bfs_sharded_strong"]["timeline"] = timeline_of(comms["shared"], STRONG_BUDGET)
        out["bfs_sharded_strong"]["note"] = "the curve to read across N: one 4e8-node search, >= 5e7 nodes per rank at 8 ranks"
    except Exception as e:  # noqa: BLE001
        out["bfs_sharded_strong"] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1:
        # the same frontier with the budget grown with the number of GPUs (weak scaling: 1e8 nodes per GPU)
        try:
            out["bfs_sharded_weak"] = timed_sharded(comms["shared"], budget * world, reps=2)
            out["bfs_sharded_weak"]["scaling"] = "weak"
        except Exception as e:  # noqa: BLE001
            out["bfs_sharded_weak"] = {"error": f"{type(e).__name__}: {e}"}
    # BASELINE config 4 shape: bfs over the 1190 Miller-Schupp presentations; the searches are independent, so they are dealt
    # round-robin to the ranks (no data-path collective) and overlapped 16 at a time on each GPU.  A failure on one rank is
    # carried through the closing all-reduce so that no rank is left waiting.
    from ac_solver.search._common import run_search_many
    from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

    from ac_solver.search._common import run_search_groups

    groups = []
    for n in range(1, 8):  # the presentations of each n have their own max_relator_length: seven batches, all in flight together
        d = generate_miller_schupp_presentations(n, 7)
        groups.append(np.array([q for w in range(1, 8) for q in d[w]], dtype=np.int8)[rank::world])
    n_mine = sum(len(g) for g in groups)

    def sweep(kind, cyclical, published, entry, kernel, traffic_key):
        if use_dist:
            dist.barrier()
        n_solved = n_nodes = n_children = 0
        sweep_err = None
        try:
            run_search_groups(kind, groups, 10**6, cyclical)  # first call: pays for the device allocations (kept by the block pool)
        except Exception as e:  # noqa: BLE001
            sweep_err = e
        if use_dist:
            dist.barrier()
        import gc

        times = []
        for _ in range(5):  # the median of five timed sweeps (a sweep is 0.15-0.45 s; the first one behind the sharded searches runs up to 40 % longer), no garbage collection while the clock runs
            n_solved = n_nodes = n_children = 0
            gc.collect()
            gc.disable()
            t0 = time.perf_counter()
            try:
                if sweep_err is None:
                    for res in run_search_groups(kind, groups, 10**6, cyclical):
                        for ok, _, s1 in res:
                            n_solved += ok
                            n_nodes += s1["nodes"]
                            n_children += s1["children"]
            except Exception as e:  # noqa: BLE001
                sweep_err = e
            tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            gc.enable()
            if use_dist:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            times.append(float(tmax[0]))
        tot = torch.tensor([n_solved, n_nodes, n_mine, 0.0 if sweep_err is None else 1.0, n_children], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tot)
        dt1 = sorted(times)[len(times) // 2]
        if float(tot[3]) > 0:
            return {"error": f"{type(sweep_err).__name__}: {sweep_err}" if sweep_err is not None else "failed on another rank"}
        # (64 + 72 f) B per generated child over the whole sweep (the seven widths together), per GPU; wall time of the
        # sweep (the searches of a group run inside ONE launch, so there is no per-search device time to sum)
        roof = search_roofline({"children": int(tot[4]), "nodes": int(tot[1]), "seconds": dt1 * world}, kernel, traffic_key)
        roof["unit"] = "GB/s per GPU"
        roof["timed"] = "wall clock of the whole sweep (median of five), host-side result handling included"
        return {"searches": int(tot[2]), "budget": 10**6, "cyclical": cyclical, "solved": int(tot[0]), "published_solved": published,
                "nodes": int(tot[1]), "children": int(tot[4]), "seconds": dt1, "samples_seconds": times, "nodes_per_s": float(tot[1]) / dt1,
                "children_per_s": float(tot[4]) / dt1, "searches_per_s": float(tot[2]) / dt1,
                "n_gpus": world, "scaling": "strong", "entry": entry, "roofline": roof}

    out["bfs_ms_sweep"] = sweep(_acx.SEARCH_BFS, True, 278,
                                "acx_search_many per rank: the 170 searches of a max_relator_length share the launches of the fused single search "
                                "(acx_bfs_many.h: a batch of every running search per round of four launches), the seven lengths one after "
                                "another; searches dealt round-robin to the ranks",
                                "k_bfs_expand_insert_many<u64> / <u128> + k_bfs_count_many + k_bfs_compact_many (a tile of one search's batch per workgroup)",
                                "bfs_ms_sweep_1e6")
    # the reference's other published experiment: greedy_search, budget 1e6, on the same 1190 (533 solved)
    out["greedy_ms_sweep"] = sweep(_acx.SEARCH_GREEDY, False, 533,
                                   "acx_search_groups per rank: the searches of all seven max_relator_lengths are jobs that a fixed set of persistent "
                                   "workgroups (one per compute unit, shared out between the 64- and the 128-bit launch) takes from a counter "
                                   "(k_greedy_sched); searches dealt round-robin to the ranks",
                                   "k_greedy_sched<u64> / <u128> (persistent workgroups, one search after the other)", "greedy_ms_sweep_1e6")
    if world == 1 and not use_dist:
        for kind, name in ((_acx.SEARCH_BFS, "bfs"), (_acx.SEARCH_GREEDY, "greedy_search")):
            b = budget if kind == _acx.SEARCH_BFS else min(budget, 10**7)
            t0 = time.perf_counter()
            run_search(kind, p, b, False)  # first call: also pays for the device allocations (kept by the block pool afterwards)
            first = time.perf_counter() - t0
            runs = []
            for _ in range(3):  # the median of three timed searches
                t0 = time.perf_counter()
                ok, path, s1 = run_search(kind, p, b, False)
                runs.append((time.perf_counter() - t0, s1))
            dt1, s1 = sorted(runs, key=lambda r: r[0])[1]
            out[name] = {"nodes_per_s": s1["nodes"] / dt1, "nodes": s1["nodes"], "seconds": dt1, "device_seconds": s1["seconds"],
                         "first_call_seconds": first, "batches": s1["levels"], "entry": "acx_search",
                         "roofline": search_roofline(s1, "k_bfs_expand_insert<u64> (expand + visited-table dedup, one launch per batch)"
                                                     if kind == _acx.SEARCH_BFS else "k_greedy_persistent<u64> (one workgroup) + the whole-GPU k_gm_* cycle for buckets of >= 512 parents, chained on the stream",
                                                     "bfs_ak3_1e8" if kind == _acx.SEARCH_BFS else "greedy_ak3_1e7")}
    if use_dist:
        # LAST (everything above is already recorded should this stage hang): the mask all-reduce on a communicator of its own.
        # torch's NCCL backend runs a communicator's collectives in issue order on one internal stream, and the orchestrator issues
        # all-to-all(k + 1) before all-reduce(k): on the shared communicator commit(k) waits for the exchange of chunk k + 1.
        try:
            comms["own"] = TorchDistComm(dev, mask_group="own")
            own = {}
            for name, b in (("bfs_sharded", budget), ("bfs_sharded_strong", STRONG_BUDGET)):
                own[name] = timed_sharded(comms["own"], b)
                own[name]["timeline"] = timeline_of(comms["own"], b)
                own[name]["mask_all_reduce_group"] = "own"
                shared = out.get(name, {})
                if "nodes_per_s" in shared:
                    shared["by_mask_group"] = {"shared": {"seconds": shared["seconds"], "nodes_per_s": shared["nodes_per_s"], "timeline": shared.get("timeline")},
                                               "own": {"seconds": own[name]["seconds"], "nodes_per_s": own[name]["nodes_per_s"], "timeline": own[name]["timeline"]}}
                    if own[name]["nodes_per_s"] > shared["nodes_per_s"]:  # the line's figure is the better of the two, and says which
                        for k in ("nodes_per_s", "seconds", "samples_seconds", "samples_setup_loop_seconds", "roofline", "timeline", "collectives"):
                            shared[k] = own[name][k]
                        shared["mask_all_reduce_group"] = "own"
        except Exception as e:  # noqa: BLE001
            out["mask_group_own"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def cpu_search_baseline(budget=10**6):
    from oracle import ac_oracle as O

    p = ak3_at_L()
    res = {}
    for fn, name in ((O.bfs, "bfs"), (O.greedy_search, "greedy_search")):
        t0 = time.perf_counter()
        ok, path, st = fn(p, budget, stats=True)
        res[name + "_nodes_per_s"] = st["nodes"] / (time.perf_counter() - t0)
    res["search_sample"] = f"AK(3) at L=25, budget {budget}, oracle/ac_oracle.c, 1 core"
    return res


def cpu_python_baseline(states, budget_s=4.0):
    """BASELINE.md section 4, leg (3): the pure Python / NumPy restatement that mirrors the reference call for call
    (oracle/ac_numpy.py, pinned on the reference's fixtures) -- what a user of the reference gets from one interpreter."""
    from oracle import ac_numpy as P

    rng = np.random.default_rng(0)
    envs = [P.Env(states[i], HORIZON) for i in range(16)]
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for e in envs:
            for a in rng.integers(0, 12, size=64):
                _, _, done, trunc = e.step(int(a))
                if done or trunc:
                    e.reset()
            steps += 64
    dt = time.perf_counter() - t0
    out = {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{steps} ACEnv.step calls (16 envs, MS initial states, L={L}) by oracle/ac_numpy.py (NumPy, one interpreter) in {dt:.1f} s"}
    p = ak3_at_L()
    for fn, name, b in ((P.bfs, "bfs", 3000), (P.greedy_search, "greedy_search", 3000)):
        t0 = time.perf_counter()
        fn(p, b)
        out[name + "_nodes_per_s"] = b / (time.perf_counter() - t0)
    out["search_sample"] = "AK(3) at L=25, budget 3000, oracle/ac_numpy.py"
    return out


def extra_env_numbers(dev, pool):
    """Context for the headline: the same env kernel where HBM, not launch latency, is the limit (4 Mi envs), and the
    fused T-step rollout kernel (state resident in registers, reward/done per step) on the config-2 batch."""
    import torch

    from ac_solver import _acx
    from ac_solver.envs.vec_env import ACVecEnv

    out = {}
    # the reference's own single-call surface, unchanged user code: ACEnv.step on ONE environment and ACMove on one presentation
    # (reference in the build container, BASELINE.md: 105 us per step, 34 us per ACMove; the intended route here is ACVecEnv)
    try:
        from ac_solver.envs.ac_env import ACEnv, ACEnvConfig
        from ac_solver.envs.ac_moves import ACMove

        e1 = ACEnv(ACEnvConfig(initial_state=pool[600], horizon_length=HORIZON))
        acts = np.random.default_rng(0).integers(0, 12, size=400)
        for a in acts[:50]:
            e1.step(int(a))
        t0 = time.perf_counter()
        for a in acts[50:]:
            e1.step(int(a))
        step_us = (time.perf_counter() - t0) / 350 * 1e6
        st, ln = pool[600].copy(), None
        for a in acts[:50]:
            st, ln = ACMove(int(a), st, L, ln)
        t0 = time.perf_counter()
        for a in acts[50:]:
            st, ln = ACMove(int(a), st, L, ln)
        move_us = (time.perf_counter() - t0) / 350 * 1e6
        out["single_call_surface"] = {"single_env_step_us": step_us, "acmove_call_us": move_us, "reference_step_us": 105.0, "reference_acmove_us": 34.0,
                                      "note": "one environment per call: one launch on the pinned staging block and one synchronisation (acx_env_step_host / "
                                              "acx_move_batch); reference figures measured in the build container (BASELINE.md)"}
        del e1
    except Exception as e:  # noqa: BLE001
        out["single_call_surface"] = {"error": f"{type(e).__name__}: {e}"}
    # The same kernel where launch overhead is < 5 % of a launch: 2^20 and 4 Mi envs, eager launches back to back, outputs rotating
    # over 8 rollout rows.  profiles/r3_env_step_roofline.json holds the rocprofv3 --kernel-trace average and the FETCH_SIZE /
    # WRITE_SIZE passes of the same command (tools/profile_env_r3.sh): trace and HIP events agree to 1-4 % at these sizes.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    regimes = {}
    for n_big in (1 << 20, 1 << 22):
        rows, k_tape, k_block = 8, 44, 40
        env = ACVecEnv(pool[np.arange(n_big) % len(pool)], horizon_length=HORIZON, record_actions=False, final_info=False)
        tape = torch.randint(0, 12, (k_tape, n_big), dtype=torch.uint8, device=dev)
        obs = torch.empty((rows, n_big, 2 * L), dtype=torch.int8, device=dev)
        rew = torch.empty((rows, n_big), dtype=torch.float32, device=dev)
        done = torch.empty((rows, n_big), dtype=torch.bool, device=dev)
        trunc = torch.empty((rows, n_big), dtype=torch.bool, device=dev)

        def step(k):
            r = k % rows
            _acx.check(_acx.lib.acx_env_step(env._h.ptr, tape[k % k_tape].data_ptr(), _acx.U8, obs[r].data_ptr(), _acx.I8, rew[r].data_ptr(), 0.0, 0.0,
                                             done[r].data_ptr(), trunc[r].data_ptr(), None, 1, env._stream()))

        # Blocks of 40 launches back to back for >= 0.3 s of device time: the FIRST block is a burst right after lighter work (what
        # rounds 1-3 reported: 67-69 us per 4 Mi-env launch), the LAST ones are the sustained rate of a kernel that keeps HBM
        # saturated (78-79 us: the same figure tools/env_roofline.py measures in a process of its own, by HIP events and by
        # rocprofv3 --kernel-trace -- profiles/r6_env_step_roofline.json: 74 us by HIP events, 69 us traced on this round's kernel).  `us_per_step_launch` is the sustained median.
        k = 0
        for _ in range(4):
            step(k)
            k += 1
        blocks, t_begin = [], time.perf_counter()
        while (time.perf_counter() - t_begin < 0.3 or len(blocks) < 5) and len(blocks) < 400:
            e0.record()
            for _ in range(k_block):
                step(k)
                k += 1
            e1.record()
            torch.cuda.synchronize()
            blocks.append(e0.elapsed_time(e1) * 1e3 / k_block)
        tail = sorted(blocks[-max(5, len(blocks) // 4):])
        us = tail[len(tail) // 2]
        # What of a launch's bytes can be served by the 256 MiB Infinity Cache (MI355X_MICROARCH.md: a buffer stays resident only while
        # it plus every byte moved between two uses of it fits): the packed state (48 B per env-step, reused every launch) does while
        # one launch moves < 256 MiB in all; the action / observation / reward / flag streams (57 B) never do (a row comes back
        # after 8 launches).  The FETCH_SIZE / WRITE_SIZE counters cannot tell the two apart (they count Infinity-Cache hits).
        per_launch = (ALGO_BYTES_PER_STEP - 2) * n_big  # bytes the kernel really moves: 24 + 24 + 1 + 50 + 4 + 1 + 1 = 105 B per env
        state_resident = per_launch < (256 << 20)
        beyond = (57 if state_resident else 105) * n_big
        regimes[str(n_big)] = {"envs": n_big, "us_per_step_launch": us, "us_per_step_launch_first_block": blocks[0], "blocks_timed": len(blocks), "launches_per_block": k_block,
                               "env_steps_per_s": n_big / us * 1e6,
                               "achieved_GBps_algorithmic": ALGO_BYTES_PER_STEP * n_big / us / 1e3,
                               "frac_of_hbm_peak_algorithmic": ALGO_BYTES_PER_STEP * n_big / us / 1e3 / HBM_PEAK_GBS,
                               "hbm_bytes_beyond_mall_per_launch": beyond, "frac_of_hbm_peak_beyond_mall": beyond / us / 1e3 / HBM_PEAK_GBS,
                               "state_stays_in_infinity_cache": state_resident}
        tracked = os.path.join(ROOT, "profiles", "r6_env_step_roofline.json")
        if os.path.exists(tracked):  # the same kernel and batch in a process of its own (tools/profile_env_r6.sh): HIP events and the kernel trace of the same command
            with open(tracked) as fh:
                t = json.load(fh).get(str(n_big), {})
            if t.get("rocprof_kernel_trace"):
                regimes[str(n_big)]["tracked"] = {"hip_event_us": t["hip_event"]["hip_event_us_per_launch"], "rocprof_avg_us": t["rocprof_kernel_trace"]["avg_us"],
                                                  "rocprof_median_us": t["rocprof_kernel_trace"]["median_us"], "traffic_bytes_per_launch": t.get("traffic_bytes_per_launch"),
                                                  "source": f"profiles/r6_env_step_roofline.json + r6_env_step_{n_big}_int8_kernel_stats.csv"}
        del env, obs, tape, rew, done, trunc
    out["throughput_regime"] = regimes["4194304"]
    out["throughput_regime"]["hbm_honest"] = True   # THE size to read the env kernel's HBM fraction from
    out["throughput_regime"]["note"] = ("4 Mi envs: one launch moves 440 MB, more than the 256 MiB Infinity Cache, so state and streams all come from / go to HBM; "
                                        "algorithmic = SURVEY 8(d)'s 107 B per env-step (the kernel moves 105: its state is 2 x 24 B packed, not 2 x 25)")
    out["throughput_regime_1Mi"] = regimes["1048576"]
    out["throughput_regime_1Mi"]["hbm_honest"] = False
    out["throughput_regime_1Mi"]["note"] = ("2^20 envs: one launch moves 110 MB; the 50 MB of packed state stay in the Infinity Cache, the 57 B per env-step of "
                                            "action / observation / reward / flag streams are what reaches HBM: read frac_of_hbm_peak_beyond_mall, not the algorithmic fraction")
    n, T = N_ENVS, 1000
    env = ACVecEnv(pool[np.arange(n) % len(pool)], horizon_length=HORIZON, record_actions=False, final_info=False)
    tape = torch.randint(0, 12, (T, n), dtype=torch.uint8, device=dev)
    rw = torch.empty((T, n), dtype=torch.float32, device=dev)
    dn = torch.empty((T, n), dtype=torch.bool, device=dev)
    tr = torch.empty((T, n), dtype=torch.bool, device=dev)
    env.rollout(tape, rw, dn, tr)
    torch.cuda.synchronize()
    e0.record()
    env.rollout(tape, rw, dn, tr)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    out["fused_rollout"] = {"envs": n, "steps": T, "env_steps_per_s": n * T / ms * 1e3, "us_per_step": ms * 1e3 / T,
                            "outputs": "f32 reward + done + truncated per step; state stays in registers (7 B/step algorithmic)"}
    del env, tape, rw, dn, tr
    # BASELINE config 5 shape on one GPU: PPO rollout with the reference's 2 x 256 tanh policy sampling on the device, float32
    # observations / rewards / terminated flags written by the env kernel straight into the [T, N, ...] rollout tensors
    try:
        from types import SimpleNamespace

        from ac_solver.agents.ppo_agent import Agent

        n, T = 1 << 17, 32  # 2^20 envs over 8 GPUs = 131 072 per GPU
        env = ACVecEnv(pool[np.arange(n) % len(pool)], horizon_length=HORIZON, obs_dtype="float32", clip_rewards=(-10, 1000), record_actions=False,
                       final_info=False)
        agent = Agent(SimpleNamespace(single_observation_space=SimpleNamespace(shape=(2 * L,)), single_action_space=SimpleNamespace(n=12)), [256, 256]).to(dev)
        obs = torch.zeros((T + 1, n, 2 * L), device=dev)
        rew = torch.zeros((T, n), device=dev)
        term = torch.zeros((T + 1, n), dtype=torch.bool, device=dev)
        trunc = torch.zeros(n, dtype=torch.bool, device=dev)
        obs[0].copy_(env.reset()[0])

        from ac_solver.agents.fused_policy import FusedPolicy

        fused = FusedPolicy(agent, 2 * L)
        act = torch.zeros((T, n), dtype=torch.int64, device=dev)
        logp = torch.zeros((T, n), device=dev)
        val = torch.zeros((T, n), device=dev)

        def rollout(policy):
            for t in range(T):
                if policy == "fused":      # acx_policy_sample: both MLPs + the action draw in one MFMA kernel (bf16 operands, f32 accumulation)
                    fused.sample(obs[t], act[t], logp[t], val[t])
                    action = act[t]
                elif policy == "torch":    # the reference's f32 torch modules
                    with torch.no_grad():
                        action = agent.get_action_and_value(obs[t])[0]
                else:
                    action = tape8[t]
                env.step(action, out=(obs[t + 1], rew[t], term[t + 1], trunc), check_errors=False)

        tape8 = torch.randint(0, 12, (T, n), dtype=torch.uint8, device=dev)
        res = {}
        def timed(fn, *a, reps=5):
            """median of `reps` timed passes behind a warm-up pass (one 32-step window is a few ms: a single pass is noisy)"""
            fn(*a)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0.record()
                fn(*a)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2]

        for name in ("fused", "torch", "none"):
            res[name] = n * T / timed(rollout, name, reps=3 if name == "torch" else 5) * 1e3
        # the same rollout with int8 observation rows (a quarter of the bytes the env kernel writes and the policy kernel reads;
        # the update would widen a minibatch to f32 when it needs it)
        env8 = ACVecEnv(pool[np.arange(n) % len(pool)], horizon_length=HORIZON, obs_dtype="int8", clip_rewards=(-10, 1000), record_actions=False,
                        final_info=False)
        obs8 = torch.zeros((T + 1, n, 2 * L), dtype=torch.int8, device=dev)
        obs8[0].copy_(env8.reset()[0])

        def rollout8():
            for t in range(T):
                fused.sample(obs8[t], act[t], logp[t], val[t])
                env8.step(act[t], out=(obs8[t + 1], rew[t], term[t + 1], trunc), check_errors=False)

        res["fused_i8"] = n * T / timed(rollout8) * 1e3
        flop = 2.0 * ((2 * L) * 256 + 256 * 256 + 256 * 12) + 2.0 * ((2 * L) * 256 + 256 * 256 + 256)  # actor + critic, per environment
        out["ppo_rollout"] = {"envs": n, "steps": T, "env_steps_per_s": res["fused"], "env_steps_per_s_torch_f32_policy": res["torch"],
                              "env_steps_per_s_env_kernel_only": res["none"], "env_steps_per_s_int8_obs": res["fused_i8"],
                              "policy": "50-256-256-12 / 50-256-256-1 tanh MLPs; fused = acx_policy_sample (v_mfma_f32_32x32x16_bf16, f32 accumulation, "
                                        "Gumbel-max draw in the kernel), torch = the reference's f32 modules",
                              "policy_tflops_fused": flop * n / max(n / res["fused"] - n / res["none"], 1e-9) / 1e12,
                              "timing": "median of 5 passes of 32 steps behind a warm-up pass (3 for the torch policy)",
                              "obs": "float32 [T+1, N, 50] written by the env kernel", "algorithmic_GBps_env_only": (12 * L + 10) * res["none"] / 1e9}
    except Exception as e:
        out["ppo_rollout"] = {"error": f"{type(e).__name__}: {e}"}
    # PPO training end to end at the same shape: rollout of 32 steps x 131 072 envs with the fused policy kernel,
    # curriculum bookkeeping, f32 behaviour statistics, GAE and the f32 torch update (4 minibatches of 1 Mi samples) per update
    if not os.environ.get("WORLD_SIZE"):  # (train_ppo brings its own process group up under a launcher)
        try:
            out["ppo_train"] = ppo_train_number()
        except BaseException as e:  # noqa: BLE001  (SystemExit included: the line must survive)
            out["ppo_train"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def ppo_train_number(n=1 << 17, T=32, updates=8):
    """ms per PPO update of ac_solver.agents.ppo.train_ppo at BASELINE config 5's per-GPU shape: the difference of a run of 2 + `updates`
    updates and a run of 2 (the smaller of two samples each: a call's fixed cost -- files, allocations -- varies), behind a warm-up run"""
    import contextlib
    import tempfile

    import torch

    from ac_solver.agents.ppo import train_ppo

    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())
    try:
        def run(u):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(sys.stderr):  # (stdout carries exactly one JSON line)
                train_ppo(["--num-envs", str(n), "--num-steps", str(T), "--total-timesteps", str(u * T * n), "--tile-initial-states", "--fused-policy",
                           "--horizon-length", "200", "--num-minibatches", "4"])
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        run(2)
        a, b = min(run(2), run(2)), min(run(2 + updates), run(2 + updates))
        per = (b - a) / updates
        return {"envs": n, "steps_per_update": T, "ms_per_update": per * 1e3, "env_steps_per_s": n * T / per, "updates_timed": updates,
                "what": "rollout with acx_policy_sample on int8 observation rows + curriculum bookkeeping + f32 behaviour statistics + GAE + the f32 torch "
                        "update (4 minibatches of 1 Mi samples, split-K weight gradients), one GPU",
                "round3": {"ms_per_update": 212.0}}
    finally:
        os.chdir(cwd)


def launch_ranks(n, argv, child=None, env=None, grace_s=20.0):
    """`python bench.py --gpus N` outside a launcher: start N FRESH rank processes (one per GPU) and relay rank 0's one JSON line.

    This process never imports torch and never touches HIP: it only spawns children -- a process that has initialised the GPU must
    not be replaced or forked on this pool.  Each child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free
    MASTER_PORT, exactly what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (the driver's launcher) would export,
    and runs this same file with the same arguments.  Rank 0's stdout is passed through; the other ranks' stdout goes to stderr.
    Returns the exit code: 0 only if EVERY rank exited 0; when one rank fails the others are given `grace_s` seconds (their
    collectives are gone) and are then terminated by PID.  `child` (tests): the command to run instead of this file."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = list(child) if child is not None else [sys.executable, os.path.abspath(__file__)]
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ACX_BENCH_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on these hosts
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd + list(argv), env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading

    lines = []

    def pump():
        for raw in procs[0].stdout:
            lines.append(raw.decode("utf-8", "replace"))

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    codes = [None] * n
    failed_at = None
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
                if codes[r] not in (None, 0) and failed_at is None:
                    failed_at = time.monotonic()
                    print(f"[bench] rank {r} exited with code {codes[r]}; waiting {grace_s:.0f} s for the other ranks", file=sys.stderr, flush=True)
        if failed_at is not None and time.monotonic() - failed_at > grace_s:
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.terminate()  # by PID: the children this call started, nothing else
                    try:
                        codes[r] = pr.wait(10)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
        time.sleep(0.05)
    th.join(5)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    if bad:
        print(f"[bench] --gpus {n}: rank(s) {', '.join(f'{r} (exit {c})' for r, c in bad)} failed; no result line is relayed", file=sys.stderr, flush=True)
        return next((c for _, c in bad if c and c > 0), 1)
    if len(json_lines) != 1:
        print(f"[bench] --gpus {n}: rank 0 printed {len(json_lines)} JSON lines instead of one", file=sys.stderr, flush=True)
        return 1
    got = json.loads(json_lines[0])
    if got.get("n_gpus") != n:
        print(f"[bench] --gpus {n}: rank 0 reports n_gpus = {got.get('n_gpus')}", file=sys.stderr, flush=True)
        return 1
    sys.stdout.write(json_lines[0] if json_lines[0].endswith("\n") else json_lines[0] + "\n")
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one process per GPU).  Under a launcher (WORLD_SIZE set) it must agree with it; "
                                                           "without one and N > 1, bench.py starts the N rank processes itself")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=N_ENVS, help="envs per GPU (default: BASELINE config 2)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: the K launches are captured once into a hipGraph and replayed; eager: K host launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true", help="skip the BFS / greedy frontier numbers")
    ap.add_argument("--search-budget", type=int, default=10**8)
    ap.add_argument("--no-extras", action="store_true", help="skip the 4 Mi-env and fused-rollout context numbers")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        # no launcher around us: become one -- BEFORE torch is imported or any HIP call is made in this process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and args.gpus is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher exported WORLD_SIZE={env_world}: refusing to report a mislabelled line")

    import torch
    import torch.distributed as dist

    world = int(env_world or "1")
    use_dist = world > 1 or bool(os.environ.get("ACX_BENCH_FORCE_DIST"))  # the env var drives the N > 1 code path on one GPU (testing)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()  # (counting does not initialise the GPU)
    if n_dev <= local or n_dev < int(os.environ.get("LOCAL_WORLD_SIZE", "1")):  # (torchrun and launch_ranks both export LOCAL_WORLD_SIZE)
        # one process per GPU: N ranks need N devices -- never N ranks sharing a device under an `n_gpus: N` label
        raise SystemExit(f"bench.py: rank {rank} of {world} needs GPU {local}, but this node shows {n_dev} device(s); --gpus {world} cannot be measured here")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the AC hot path only exists as HIP kernels (no CPU fallback)")
    torch.cuda.set_device(local)

    from ac_solver import _acx
    from ac_solver.envs.vec_env import ACVecEnv

    K, W, N = args.steps, args.warmup, args.envs
    pool = ms_pool_at_L(L)
    states = pool[(np.arange(N) + rank * N) % len(pool)]
    env = ACVecEnv(states, horizon_length=HORIZON, obs_dtype="int8", record_actions=False, final_info=False)
    dev = env.device
    T = K + W
    tape = torch.as_tensor(np.random.default_rng(rank).integers(0, 12, size=(T, N), dtype=np.uint8), device=dev)
    # PPO-style rollout buffers, resident in HBM before the timed region: ROWS rows, step k of pass p writes row (p K + k) mod ROWS
    ROWS = max(K, MIN_ROLLOUT_ROWS)
    obs = torch.empty((ROWS, N, 2 * L), dtype=torch.int8, device=dev)
    rew = torch.zeros((ROWS, N), dtype=torch.float32, device=dev)
    done = torch.empty((ROWS, N), dtype=torch.bool, device=dev)
    trunc = torch.empty((ROWS, N), dtype=torch.bool, device=dev)

    def launch(t, slot):
        _acx.check(_acx.lib.acx_env_step(env._h.ptr, tape[t].data_ptr(), _acx.U8, obs[slot].data_ptr(), _acx.I8, rew[slot].data_ptr(), 0.0, 0.0,
                                         done[slot].data_ptr(), trunc[slot].data_ptr(), None, 1, env._stream()), "acx_env_step")

    env.reset()
    for t in range(W):  # untimed warm-up steps
        launch(t, t % ROWS)
    torch.cuda.synchronize()

    R = max(1, -(-MIN_TIMED_STEPS // K))  # passes over the K steps in the timed window; same on every rank (depends on K only)
    P = max(1, min(R, GRAPH_NODES_MAX // K))  # passes per graph
    G = -(-R // P)                            # graph launches in the timed window
    R = P * G
    mode = args.mode
    graph = None
    if mode == "graph":
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):  # capturing does not execute: the env stays in its post-warm-up state
                for p_ in range(P):
                    for k in range(K):
                        launch(W + k, (p_ * K + k) % ROWS)
        except Exception as e:  # pragma: no cover
            print(f"[bench] graph capture failed ({e}); falling back to eager launches", file=sys.stderr)
            graph, mode = None, "eager"
    torch.cuda.synchronize()

    # the process group is created only now: the RCCL watchdog thread must not touch the runtime while the
    # graph above is being captured (envs are independent, so nothing before this point communicates)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL writes its version banner (NCCL_DEBUG=VERSION on these boxes) to the C stdout; stdout must carry exactly one
        # JSON line, so file descriptor 1 points at stderr while the communicator comes up
        import ctypes

        libc = ctypes.CDLL(None)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            libc.fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        if graph is not None:
            graph.replay()  # one untimed replay after the communicator came up
            torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run_k_steps():  # P passes of the K steps
        if graph is not None:
            graph.replay()
        else:
            for p_ in range(P):
                for k in range(K):
                    launch(W + k, (p_ * K + k) % ROWS)

    # one isolated graph launch (P passes; round 1 timed one pass): carries the fixed cost of one graph replay + synchronize
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_k_steps()
    torch.cuda.synchronize()
    single_wall = time.perf_counter() - t0
    # the timed region: R back-to-back passes of the same K steps (the env keeps stepping, rows of the rollout buffers are
    # rewritten), barrier + synchronize on both sides, HIP events on the launch stream around the same window
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(G):
        run_k_steps()
    ev1.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream: R * K back-to-back env kernels

    tmax = torch.tensor([wall, dev_ms, single_wall], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall, dev_ms, single_wall = float(tmax[0]), float(tmax[1]), float(tmax[2])

    # sanity: the timed steps really ran (count_steps advanced, rewards written)
    written = min(ROWS, P * K)
    assert int(env.get_counts().max()) > 0 and bool(torch.isfinite(rew[:written]).all()) and bool((rew[:written] != 0).all())

    def headline(extras, search):
        total_steps = N * K * R * world
        launch_s = dev_ms * 1e-3 / (K * R)
        achieved = ALGO_BYTES_PER_STEP * N / launch_s / 1e9
        out = {
            "metric": "AC env-steps/sec at max_relator_len=25",
            "value": total_steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": wall * 1e3 / (K * R),
            "repeats": R,
            "timed_window_ms": wall * 1e3,
            "graph_nodes": K * P if mode == "graph" else 0,
            "ms_per_step_single_pass": single_wall * 1e3 / (K * P),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{N} batched ACEnv.step per GPU, Miller-Schupp initial states, max_relator_len={L}, horizon {HORIZON}, "
                                   "random action tape, int8 obs + f32 reward + done/truncated into HBM rollout buffers",
                       "envs_per_gpu": N, "max_relator_length": L, "launch": mode, "state": "int8 letters packed 2 bit per letter in one u64 per relator",
                       "parallelism": f"dp{world} (independent envs, no collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_env_step<u64,int8>", "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP * N,
                         "avg_launch_us": launch_s * 1e6, "launches_timed": K * R},
        }
        traffic_file = os.path.join(ROOT, "profiles", "env_step_traffic.json")
        if os.path.exists(traffic_file) and N == N_ENVS:  # HBM bytes per launch from rocprofv3 --pmc passes (see profiles/README.md)
            with open(traffic_file) as f:
                out["roofline"]["traffic"] = json.load(f).get("hbm_bytes_per_launch")
            # not measured in this run: counters need their own rocprofv3 passes (tools/profile_env_r3.sh; 2 x FETCH_SIZE + WRITE_SIZE)
            out["roofline"]["traffic_source"] = "profiles/env_step_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same kernel and batch)"
        # the same kernel under rocprofv3 --kernel-trace: the profiler stamps every dispatch of the replayed graph begin-to-end
        # (launch latency included, no overlap with its neighbours), which is longer than the back-to-back launch period timed here
        # At this batch size the K rollout rows + the state (K x 3.7 MB + 1.6 MB) sit in the 256 MiB Infinity Cache whenever K <= ~60:
        # nothing of a launch HAS to reach HBM before the next one starts -- the launch is bound by latency (kernel boundary + one
        # dependent memory round trip + instruction issue), and `frac` says how far that is from the HBM roofline, not which
        # resource is saturated.
        rows_used = min(ROWS, P * K)
        working_set = rows_used * (2 * L + 6) * N + 24 * N
        out["roofline"]["hbm_bytes_beyond_mall"] = 0 if working_set < (256 << 20) else (ALGO_BYTES_PER_STEP - 50) * N
        out["roofline"]["working_set_bytes"] = working_set
        out["roofline"]["rollout_rows"] = rows_used
        # the fractions by name: by the launch PERIOD timed in this run (= `frac`), and counting only the bytes that HAVE to
        # reach HBM at this batch size (the packed state, 24 B per environment, stays in the 256 MiB Infinity Cache from launch to launch)
        out["roofline"]["frac_by_period"] = achieved / HBM_PEAK_GBS
        out["roofline"]["frac_hbm_beyond_mall"] = out["roofline"]["hbm_bytes_beyond_mall"] / launch_s / 1e9 / HBM_PEAK_GBS
        ev = os.path.join(ROOT, "profiles", "r6_env_step_roofline.json")
        if os.path.exists(ev) and N == N_ENVS:
            with open(ev) as f:
                r6 = json.load(f)
            here = r6.get(str(N), {})
            kt, stp = here.get("rocprof_kernel_trace"), r6.get("stamps_65536")
            if kt:
                # the same command under rocprofv3 --kernel-trace: begin-to-end of every dispatch of the replayed graph.  A dispatch that
                # is longer than the launch PERIOD measured here cannot be the same thing (an isolated dispatch's latency): no fraction then.
                rp = {"avg_kernel_us": kt["avg_us"], "min_kernel_us": kt["min_us"], "calls": kt["calls"], "period_us_under_rocprof": kt.get("period_us_under_rocprof"),
                      "source": "profiles/r6_env_step_65536_kernel_stats.csv (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-search --no-extras "
                                "--no-cpu-baseline: tools/profile_env_r6.sh)"}
                if kt["avg_us"] <= launch_s * 1e6:
                    rp["frac_by_kernel_duration"] = ALGO_BYTES_PER_STEP * N / (kt["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
                else:
                    rp["note"] = ("the traced dispatches are longer than the launch period timed in this run (the profiler serialises the graph's nodes): not comparable, no "
                                  "fraction derived; the traces that DO agree with their HIP events are at_1Mi_envs / at_4Mi_envs / f32_obs_131072_envs below")
                out["roofline"]["rocprof"] = rp
            busy = here.get("sq_busy")
            if busy:  # profiler-side busy time per launch: SQ_BUSY_CYCLES calibrated on the 4 Mi-env launch, whose duration trace and events agree on
                out["roofline"]["profiler_busy"] = dict(busy, source="profiles/r6_env_step_roofline.json (rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/env_roofline.py)")
            elif here.get("GRBM_GUI_ACTIVE_per_launch"):
                out["roofline"]["profiler_busy"] = {"GRBM_GUI_ACTIVE_per_launch": here["GRBM_GUI_ACTIVE_per_launch"], "busy_us_at_2p4GHz": here["grbm_busy_us_per_launch_at_2p4GHz"],
                                                    "note": "under --pmc every dispatch runs alone (51 us apart): GRBM_GUI_ACTIVE then spans the profiler's own per-dispatch work, not "
                                                            "the kernel (the guide: reads high below ~0.3 ms); at 4 Mi envs it gives 75.8 us against 68.8-74.4 us traced / timed",
                                                    "source": "profiles/r6_env_step_roofline.json"}
            if stp:  # in-kernel stamps of the diagnostic build under the same graph replay: where the period goes
                out["roofline"]["stamps"] = {"active_us": stp["active_us_median"], "launch_boundary_us": stp["gap_us_median"], "one_wave_us": stp["one_wave_us_median"],
                                             "wave_start_spread_us": stp["wave_start_spread_us_median"], "frac_by_active_time": stp["frac_of_8TBps_by_active_time"],
                                             "source": "profiles/r6_env_step_roofline.json: tools/step_stamps.py on the -DACX_STEP_STAMP build of this round's kernel (s_memrealtime per wave)"}
            tr = here.get("traffic_bytes_per_launch")
            if tr:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = ("profiles/r6_env_step_roofline.json (rocprofv3 --pmc FETCH_SIZE and, in a pass of its own, --pmc WRITE_SIZE over tools/env_roofline.py "
                                                     "65536 400 int8 128 1: same kernel, batch and row ring; 2 x FETCH_SIZE + WRITE_SIZE: gfx950 counts 128-B read requests at 64 B)")
            # where trace and HIP events agree (launch overhead < 5 %): the same kernel at 2^20 envs (state resident in the Infinity Cache: read the
            # beyond-MALL fraction), at 4 Mi envs -- the HBM-honest size, sustained (0.3 s warm-up, 5 x 40 launches) -- and with f32 observation
            # rows at BASELINE config 5's per-GPU shape; all re-taken on this round's acx_step.hip
            for key, tag, n_big, per_env, beyond in (("at_1Mi_envs", str(1 << 20), 1 << 20, ALGO_BYTES_PER_STEP, 57), ("at_4Mi_envs", str(1 << 22), 1 << 22, ALGO_BYTES_PER_STEP, 105),
                                                     ("f32_obs_131072_envs", "131072_float32", 1 << 17, 12 * L + 10, 12 * L + 10 - 50)):
                big = r6.get(tag, {})
                if not big.get("rocprof_kernel_trace"):
                    continue
                us_ev, us_tr = big["hip_event"]["hip_event_us_per_launch"], big["rocprof_kernel_trace"]["avg_us"]
                resident = key != "at_4Mi_envs"
                out["roofline"][key] = {"hip_event_us": us_ev, "rocprof_avg_us": us_tr, "frac_by_hip_events": per_env * n_big / us_ev / 1e3 / HBM_PEAK_GBS,
                                        "frac_by_rocprof_avg": per_env * n_big / us_tr / 1e3 / HBM_PEAK_GBS,
                                        "frac_of_hbm_peak_beyond_mall": beyond * n_big / us_tr / 1e3 / HBM_PEAK_GBS, "state_stays_in_infinity_cache": resident,
                                        "hbm_honest": not resident, "traffic_bytes_per_launch": big.get("traffic_bytes_per_launch"),
                                        "algorithmic_bytes_per_launch": big["algorithmic_bytes_per_launch"],
                                        "source": f"profiles/r6_env_step_{tag if '_' in tag else tag + '_int8'}_kernel_stats.csv + r6_env_step_roofline.json (tools/profile_env_r6.sh)"}
        if extras is not None:
            out["env_context"] = extras
        if search is not None:
            out["search"] = search
        return out

    # The secondary measurements of a multi-rank run go through RCCL collectives (sharded BFS).  The timed headline is
    # already in hand: if they have not come back after SEARCH_TIMEOUT_S seconds (a rank lost, a collective stuck), rank 0
    # still prints its one JSON line and every rank leaves, instead of the whole job hanging.
    watchdog = None
    partial, extras_box = {}, [None]
    if use_dist and not args.no_search:
        import threading

        def give_up():
            # the headline and every finished stage of the secondary measurements are printed; the stage that hung is named
            if rank == 0:
                got = dict(partial)
                got["error"] = f"secondary measurements did not finish within {SEARCH_TIMEOUT_S} s (finished stages: {sorted(partial)})"
                print(json.dumps(headline(extras_box[0], got)), flush=True)
            # exit code: 0 when the sharded search itself was measured (the line is usable and says what is missing), else failure
            os._exit(0 if "nodes_per_s" in partial.get("bfs_sharded", {}) else 3)

        watchdog = threading.Timer(SEARCH_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()

    extras = None
    if rank == 0 and not args.no_extras:
        try:
            extras = extra_env_numbers(dev, pool)
        except Exception as e:
            extras = {"error": f"{type(e).__name__}: {e}"}
    extras_box[0] = extras
    search = None
    if not args.no_search:
        try:
            search = search_numbers(world, rank, dev, args.search_budget, use_dist, partial)
        except Exception as e:  # the headline line must survive a failure of the secondary measurement
            search = dict(partial)
            search["error"] = f"{type(e).__name__}: {e}"
    if watchdog is not None:
        watchdog.cancel()

    # every rank is through with the GPU: the communicator goes first, so that ranks 1.. can leave while rank 0 times the CPU legs
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out = headline(extras, search)
        if not args.no_cpu_baseline:  # on every line, N = 1 or not (rank 0's host cores, after the collectives)
            out["cpu_baseline"] = cpu_baseline(states, 0, budget_s=12.0 * CPU_SCALE)
            if not args.no_search:
                out["cpu_baseline"].update(cpu_search_baseline(10**6 if CPU_SCALE == 1.0 else 10**4))
            out["cpu_baseline"]["python_numpy"] = cpu_python_baseline(states, budget_s=4.0 * CPU_SCALE)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
