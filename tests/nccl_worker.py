"""Child process of test_gpu_parity.py::test_rccl_collectives_single_rank: the path's two collectives through RCCL
(backend "nccl") with CUDA buffers, world size 1 (one GPU per box here; the 2-rank logic is covered with gloo)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from flashgmm_amd import container as Cn
from flashgmm_amd import parallel as P

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = P.all_gather_stream_lengths([11, 22, 33], 4, device=dev)
assert g.device.type == "cuda" and g.tolist() == [[11, 22, 33, -1]], g
# the overlapped form bench.py uses: one preallocated all_gather_into_tensor, asynchronous, reused
ex = P.LengthExchange(4, device=dev)
for rep in range(3):
    ex.start([11 + rep, 22, 33])
    assert ex.work is not None
    assert ex.wait().tolist() == [[11 + rep, 22, 33, -1]]
# a real 2-rank-style all_gather call on the GPU (world size 1 still goes through RCCL)
buf = torch.arange(6, dtype=torch.int64, device=dev)
out = [torch.empty_like(buf)]
dist.all_gather(out, buf)
assert torch.equal(out[0], buf)
blobs = [Cn.pack([(bytes([i] * (10 + i)), 5 + i, torch.tensor([1, 0, 1]))], (3, 4, 6)) for i in range(3)]
assert P.gather_containers(blobs, 3, device=dev) == blobs
u8 = torch.arange(200, dtype=torch.uint8, device=dev)
o8 = [torch.empty_like(u8)]
dist.all_gather(o8, u8)
assert torch.equal(o8[0], u8)
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
