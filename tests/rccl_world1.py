"""Run by tests/test_sharded.py::test_rccl_backend_world1 in a process of its own: the three multi-GPU schedules
and the collectives bench.py --gpus N uses, on the RCCL ("nccl") backend with ONE rank -- what a one-GPU box can
check of the RCCL path (the calls, dtypes and split arguments are accepted; no bytes cross a link)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import radix_sort_amd as rs
from radix_sort_amd.sharded import ShardedRadixSort
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29600 + os.getpid() % 300), rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = rs.default_context(0)
d = rs.PRIMITIVES["u32"]; n = 1 << 24
s = ShardedRadixSort()
out = torch.zeros(3, dtype=torch.int64, device="cuda")
for name, run in (("first", lambda b: s.sort_exchange_first(b, d, [n], chunks=4)), ("overlapped", lambda b: s.sort_exchange_first(b, d, [n], chunks=4, sub_ranges=4)), ("one", lambda b: s.sort_one_exchange(b, d, [n])), ("per-pass", lambda b: s.sort(b, d, [n]))):
    x = torch.empty(n * 4, dtype=torch.uint8, device="cuda")
    ctx.generate_device(x.data_ptr(), n, d, rs.GEN_UNIFORM, 5)
    run(x); torch.cuda.synchronize()
    ctx.verify_device(x.data_ptr(), n, d, out.data_ptr()); torch.cuda.synchronize()
    print(name, "descents", out[0].item(), "unstable", out[2].item(), flush=True)
    assert out[0].item() == 0 and out[2].item() == 0, name
# the collectives the N>1 path uses, on RCCL with one rank
t = torch.arange(8, dtype=torch.int64, device="cuda"); l = [torch.zeros_like(t)]
dist.all_gather(l, t); dist.all_reduce(t); dist.barrier()
a = torch.arange(16, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
dist.all_to_all_single(b, a, output_split_sizes=[16], input_split_sizes=[16])
torch.cuda.synchronize()
assert b.tolist() == list(range(16)) and l[0].tolist() == list(range(8))
print("RCCL WORLD1 OK", flush=True)
dist.destroy_process_group()
