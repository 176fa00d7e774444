import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.broadcast(t, src=0); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
out = [torch.empty_like(t)]; dist.all_gather(out, t)
torch.cuda.synchronize(); print("rccl world-1 ok", t.tolist(), dist.get_backend())
dist.destroy_process_group()
