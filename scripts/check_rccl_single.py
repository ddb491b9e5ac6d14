"""One-rank RCCL sanity check of the calls the data-parallel trainer makes: process group over "nccl",
broadcast, all-reduce of a bucket view on a side stream with event hand-off, barrier."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
flat = torch.arange(1000, dtype=torch.float32, device="cuda")
dist.broadcast(flat, src=0)
comm = torch.cuda.Stream()
ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream())
with torch.cuda.stream(comm):
    comm.wait_event(ready)
    dist.all_reduce(flat[100:600], op=dist.ReduceOp.SUM)
    done = torch.cuda.Event(); done.record(comm)
torch.cuda.current_stream().wait_event(done)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(flat.cpu(), torch.arange(1000, dtype=torch.float32))
dist.destroy_process_group()
print("rccl single-rank ok")
