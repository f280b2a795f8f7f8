#!/usr/bin/env python3
"""debug: which fail_at_call makes the ranks of acx_bfs_sharded disagree on their collectives (thread ranks on one GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np
import torch
from ac_solver.search.sharded import NativeComm, bfs_sharded_native
from tests.shard_helpers import ThreadComm, run_threads

ak3 = np.zeros(50, np.int8)
ak3[:7] = [1, 1, 1, -2, -2, -2, -2]
ak3[25:31] = [1, 2, 1, -2, -1, -2]


class LogComm(ThreadComm):
    def __init__(self, shared, rank):
        super().__init__(shared, rank)
        self.seq = []

    def all_to_all_single(self, recv, send):
        self.seq.append(("a2a", send.numel()))
        return super().all_to_all_single(recv, send)

    def all_reduce(self, t, op):
        self.seq.append((op, t.numel()))
        return super().all_reduce(t, op)


fails = [int(x) for x in sys.argv[1:]] or [2, 7, 15, 16, 17, 25, 40, 61, 90]
for fail_at in fails:
    def run(comm):
        lc = LogComm(comm.s, comm.rank)
        nat = NativeComm.from_python(lc)
        try:
            bfs_sharded_native(ak3, 30000, comm=nat, batch_parents=256, replicate_below=40, _fail_at_call=fail_at, _fail_rank=1)
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)[:200]
        return msg, lc.seq, [repr(e)[:100] for e in nat.errors]
    try:
        out = run_threads(3, run)
    except BaseException as e:
        print("fail_at", fail_at, "run_threads raised", repr(e)[:200])
        continue
    seqs = [o[1] for o in out]
    same = seqs[0] == seqs[1] == seqs[2]
    print("fail_at", fail_at, "same" if same else "DIFFERENT", [o[0] for o in out], [len(s) for s in seqs], [o[2] for o in out])
    if not same:
        n = min(len(s) for s in seqs)
        for i in range(max(len(s) for s in seqs)):
            row = [s[i] if i < len(s) else None for s in seqs]
            if i >= n or not (row[0] == row[1] == row[2]):
                print("   first difference at", i, row, "before:", seqs[0][max(0, i - 4):i])
                break
