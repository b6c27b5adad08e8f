"""Single-rank RCCL self-test (run under torch.distributed.run on the GPU box): process-group init with device_id,
barrier, all_reduce and all_to_all_single (int64 ids + float32 rows, explicit splits, sync and async) - the exact
collective calls dgsparse/dist.py issues.  World size 1 is all a 1-GPU box allows; the N>1 exchange logic is covered
on gloo by tests/test_dist_cpu.py."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
rank = int(os.environ.get('RANK', '0'))
world = int(os.environ.get('WORLD_SIZE', '1'))
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
dist.barrier()
t = torch.tensor([5], dtype=torch.int64, device=dev)
dist.all_reduce(t)
assert int(t) == 5 * world
ids = torch.arange(7, dtype=torch.int64, device=dev)
out = torch.empty_like(ids)
dist.all_to_all_single(out, ids, [7] * world if world == 1 else None, [7] * world if world == 1 else None)
assert torch.equal(out, ids)
rows = torch.rand(1000, 64, device=dev)
recv = torch.empty_like(rows)
w = dist.all_to_all_single(recv, rows, [1000], [1000], async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(recv, rows)
dist.destroy_process_group()
print('nccl selftest ok')
