import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(15_129_603, device="cuda")
ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ea.record(); dist.all_reduce(x); x.div_(1); eb.record()
gl = [torch.zeros(1, device="cuda")]
dist.all_gather(gl, torch.tensor([3.0], device="cuda"))
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
print("nccl world=1 ok", float(x[0]), float(gl[0]), float(t), "allreduce ms %.3f" % ea.elapsed_time(eb), dist.get_world_size())
dist.destroy_process_group()
