import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29517")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
t=torch.arange(8, dtype=torch.float32, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.AVG); torch.cuda.synchronize(); print("AVG ok", t.tolist())
dist.destroy_process_group()
