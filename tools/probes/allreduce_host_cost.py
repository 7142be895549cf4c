import os, time, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.zeros(1 << 20, device="cuda")
side = torch.cuda.Stream()
for _ in range(5):
    dist.all_reduce(x)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
works = []
for _ in range(n):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        works.append(dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=True))
t1 = time.perf_counter()
for w in works:
    w.wait()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"async all_reduce host cost {1e6 * (t1 - t0) / n:.1f} us per call; wait {1e6 * (t2 - t1) / n:.1f} us per call")
dist.destroy_process_group()
