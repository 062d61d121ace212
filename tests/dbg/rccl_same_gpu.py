"""Can two ranks of one RCCL communicator share a GPU on this box?  (NCCL refuses duplicate devices; asked once, here.)"""
import os, subprocess, sys
if "RANK" not in os.environ:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2")
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    rc = 0
    for p in ps:
        try:
            rc |= p.wait(timeout=100)
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 99
    print("exit codes or:", rc)
    sys.exit(rc)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=2)
x = torch.ones(4, device="cuda") * (1 + dist.get_rank())
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank", dist.get_rank(), "sum", x.tolist(), flush=True)
dist.destroy_process_group()
