#!/usr/bin/env python3
"""What RCCL says about itself on this box (one rank: all a 1-GPU box allows): version, channel count, the launch geometry of
an all-reduce of a gradient-bucket-sized buffer.  Run as
    NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,TUNING python tools/rccl_probe.py 2> gpurun_out/rccl_probe.log
The interesting lines ("channels", "nThreads", "Launch mode", "comm ... nranks") are echoed to stdout."""
import os
import re
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.ones(8 * 1024 * 1024, device="cuda:0")          # 32 MB: one gradient bucket of engine.GradReducer
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
print("nccl version", torch.cuda.nccl.version(), "all_reduce(ones) ->", float(x[0]))
dist.destroy_process_group()
