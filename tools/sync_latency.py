import os, time, torch, torch.distributed as dist
x = torch.zeros(64, device="cuda")
def lat(tag):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        x.add_(1.0)
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(200):
        x.add_(1.0)
        float(x[0])
    t2 = time.perf_counter()
    print(f"{tag}: launch+synchronize {(t1 - t0) / 200 * 1e6:.1f} us, launch+item {(t2 - t1) / 200 * 1e6:.1f} us", flush=True)
lat("before pg")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
lat("after init_process_group")
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
lat("after first collective")
dist.destroy_process_group()
lat("after destroy")
