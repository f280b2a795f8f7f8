#!/usr/bin/env python3
"""debug: which librccl does libacx resolve when torch.distributed (backend nccl) already runs one, and do its collectives work on
torch's own ncclComm_t (world 1)?"""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import torch
import torch.distributed as dist
from ac_solver import _acx
from ac_solver.search.sharded import NativeComm

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
nat = NativeComm.from_process_group()
print("rank/world", nat.rank, nat.world)
a = torch.arange(1000, dtype=torch.int64, device="cuda")
b = torch.zeros_like(a)
st = torch.cuda.current_stream().cuda_stream
print("all_to_all rc", nat.c.all_to_all(nat.c.ctx, a.data_ptr(), b.data_ptr(), a.numel(), st))
m = torch.arange(77, dtype=torch.int32, device="cuda")
print("all_reduce rc", nat.c.all_reduce(nat.c.ctx, m.data_ptr(), m.numel(), _acx.I32, _acx.RED_SUM, st))
torch.cuda.synchronize()
print("ok", bool(torch.equal(a, b)), bool(torch.equal(m.cpu(), torch.arange(77, dtype=torch.int32))))
libs = sorted({line.split()[-1] for line in open("/proc/self/maps") if "rccl" in line})
print("librccl mapped:", libs)
dist.destroy_process_group()
