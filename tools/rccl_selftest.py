"""One-rank RCCL smoke test on the GPU box: the collectives bench.py issues per level (MAX all-reduce of f32 maxima on
the device, all-gather of the f64 top-k table), through torch.distributed's nccl backend.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/rccl_selftest.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from geneticscre_amd.dist import exchange_level

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.arange(80000, dtype=torch.float32, device=dev) * (rank + 1)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
mine = torch.full((101, 5), float(rank), dtype=torch.float64, device=dev)
allrows = torch.empty((world * 101, 5), dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(allrows, mine)
dist.barrier()
torch.cuda.synchronize()
best = exchange_level(np.array([1.5, 2.5]), np.array([3, 4]), np.array([5, 6]), np.array([1, 1]), np.array([2, 2]), x[:10], 3, world, device=dev)
print("rccl ok", world, float(x[-1]), allrows.shape, best.tolist(), flush=True)
dist.destroy_process_group()
