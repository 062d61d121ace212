"""Host time of one all-to-all of small device buffers over RCCL with one rank: the public call against the backend's own."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd")]
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
inp = torch.zeros(4096, dtype=torch.uint8, device="cuda"); out = torch.zeros_like(inp)
splits = [4096]
for _ in range(20):
    dist.all_to_all_single(out, inp, splits, splits)
torch.cuda.synchronize()
def timeit(f, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("public all_to_all_single: host %.1f us per call, %.1f with the device" % timeit(lambda: dist.all_to_all_single(out, inp, splits, splits)))
pg = dist.group.WORLD._get_backend(torch.device("cuda"))
opts = dist.AllToAllOptions()
def direct():
    pg.alltoall_base(out, inp, splits, splits, opts).wait()
for _ in range(20):
    direct()
print("backend alltoall_base:    host %.1f us per call, %.1f with the device" % timeit(direct))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(20):
        direct()
    print("on a side stream:         host %.1f us per call, %.1f with the device" % timeit(direct))
dist.destroy_process_group()
