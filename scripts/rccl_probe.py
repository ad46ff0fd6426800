"""Which HIP runtime / RCCL does a process end up with, and does shz_comm_create work?  python scripts/rccl_probe.py plain | torch_first | torch_dist
(torch imported BEFORE libshz.so is loaded: one runtime, works; the other order gives two runtimes and ncclCommInitRank fails.)"""
import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    print("torch", torch.__version__, "cuda avail", torch.cuda.is_available() if mode == "x" else "(not asked)")
elif mode == "torch_dist":
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    t = torch.tensor([1.0]); dist.all_reduce(t)
from shazam_amd import _ffi
ctx = _ffi.Context(0)
try:
    comm = _ffi.Comm(ctx, _ffi.comm_unique_id(), 0, 1)
    comm.warmup(); comm.barrier()
    print(mode, "RCCL comm OK")
    comm.close()
except Exception as e:
    print(mode, "FAILED:", e)
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "rccl" in l or "amdhip" in l})
print(libs)
