"""Exercises the RCCL calls of the data-parallel path (async all-reduce with AVG, the bucketed reducer) on a one-rank
process group: what can be checked of it on a single GPU."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
torch.cuda.set_device(0)
t=torch.ones(1<<20, device="cuda")
w=dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True); w.wait(); torch.cuda.synchronize()
print("avg ok", t[0].item(), dist.get_backend())
import sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
from vit_amd.ddp import BucketedGradAllReduce
m=torch.nn.Linear(64,64).cuda()
r=BucketedGradAllReduce(m.parameters()); r.world=2  # force the launch path on a 1-rank group
m(torch.randn(8,64,device="cuda")).sum().backward(); r.finish(); torch.cuda.synchronize()
print("reducer ok", m.weight.grad.abs().sum().item()>0)
dist.barrier(); dist.destroy_process_group()
