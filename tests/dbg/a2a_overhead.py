"""Host cost of one collective through the transport on one rank (RCCL): the wrapper of torch.distributed against the
backend object called directly."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd")]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
a = torch.zeros(1024, dtype=torch.uint8, device="cuda"); b = torch.zeros_like(a)
w = torch.zeros(1, dtype=torch.int32, device="cuda")
def bench(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("all_to_all_single wrapper   host us/call, total us/call:", bench(lambda: dist.all_to_all_single(b, a, [1024], [1024])))
pg = dist.group.WORLD
be = pg._get_backend(torch.device("cuda"))
print("backend.alltoall_base        :", bench(lambda: be.alltoall_base(b, a, [1024], [1024])))
print("all_reduce wrapper           :", bench(lambda: dist.all_reduce(w, op=dist.ReduceOp.MAX)))
opts = dist.AllreduceOptions(); opts.reduceOp = dist.ReduceOp.MAX
print("backend.allreduce            :", bench(lambda: be.allreduce([w], opts)))
s = torch.cuda.Stream(priority=-1)
def on_stream():
    with torch.cuda.stream(s):
        be.alltoall_base(b, a, [1024], [1024])
print("backend.alltoall_base on side stream:", bench(on_stream))
dist.destroy_process_group()
