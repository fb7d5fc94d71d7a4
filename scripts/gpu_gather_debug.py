"""two ranks on one device, gloo control plane, CUDA tensors: generate_gathered with growing chunk sizes, printing progress"""
import os, sys, time, faulthandler, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(60, exit=True)
rank = int(os.environ["RANK"]); torch.cuda.set_device(0)
dist.init_process_group("gloo")
from pta_replicator_amd.distributed import generate_gathered
n_cols = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
def gen(n, r0, out):
    out.copy_(torch.arange(r0, r0 + n, device=out.device, dtype=out.dtype)[:, None].expand(n, out.shape[1]))
    return out
for total, chunk in ((64, 16), (512, 128), (2048, 256)):
    t0 = time.time()
    full = generate_gathered(None, total, r0=0, chunk=chunk, generate=gen, n_cols=n_cols, device=torch.device("cuda", 0))
    torch.cuda.synchronize(); dist.barrier()
    if rank == 0:
        ok = bool((full[:, 0] == torch.arange(total, device="cuda", dtype=torch.float64)).all())
        print("total", total, "chunk", chunk, "n_cols", n_cols, "ok", ok, "s", round(time.time() - t0, 2), flush=True)
os._exit(0)
