import os, torch, torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
x = torch.arange(12_000_000, dtype=torch.float32, device="cuda")
h = dist.all_reduce(x, async_op=True)
h.wait()
y = x * 2
torch.cuda.synchronize()
print("rccl world-1 async all_reduce ok", float(y[5]), dist.get_backend())
dist.barrier()
dist.destroy_process_group()
