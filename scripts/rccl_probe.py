"""Which HIP runtime / RCCL does a process end up with, and does shz_comm_create work?  python scripts/rccl_probe.py plain | torch_first | torch_dist
(torch imported BEFORE libshz.so is loaded: one runtime, works; the other order gives two runtimes and ncclCommInitRank fails.)"""
import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "scaling":
    import torch   # FIRST
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
if mode == "scaling":
    # bench.py's db_build_scaling as the driver runs it with --gpus N, but one rank: torch first, a gloo group, a real RCCL
    # communicator, exchange_run / allgather over it, torch's all_reduce for the maxima
    import json, types
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    dist.init_process_group("gloo", rank=0, world_size=1)
    import bench
    comm = _ffi.Comm(ctx, _ffi.comm_unique_id(), 0, 1)
    o = bench.db_build_scaling(types.SimpleNamespace(scaling_songs=int(sys.argv[2]) if len(sys.argv) > 2 else 2000, scaling_seconds=180.0),
                               ctx, dist, comm, 0, 1)
    print(json.dumps({k: o[k] for k in ("songs", "seconds", "fingerprint_s", "seal_and_send_s", "final_rounds_and_merge_s", "rows", "segments",
                                        "runs_sent_on_the_way_rank0", "exchange")}))
    comm.close()
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
