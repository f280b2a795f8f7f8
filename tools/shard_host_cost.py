"""Host cost of one chunk of the sharded frontier: a search in chunks so small (2^10 / 2^12 parents) that the device work is
nothing, without and with the collectives (RCCL at world 1).  DESIGN.md section 7 quotes the result."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ac-solver_amd"), ROOT]
import numpy as np, torch
from ac_solver.search.sharded import bfs_sharded
import ac_solver.search.sharded as sh
ak3 = np.zeros(50, np.int8); ak3[:7] = [1, 1, 1, -2, -2, -2, -2]; ak3[25:31] = [1, 2, 1, -2, -1, -2]
for force in (False, True):
    sh._FORCE_EXCHANGE = force
    comm = None
    if force:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        comm = sh.TorchDistComm(torch.device("cuda", 0))
    for bp in (1 << 10, 1 << 12):
        bfs_sharded(ak3, 2 * 10**6, comm=comm, batch_parents=bp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok, path, st = bfs_sharded(ak3, 2 * 10**6, comm=comm, batch_parents=bp, want_stats=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"exchange={force} chunk 2^{bp.bit_length()-1}: {st['chunks']} chunks in {dt*1e3:.1f} ms = {dt/st['chunks']*1e6:.0f} us per chunk (host-bound: tiny chunks)", flush=True)
