"""does torch.distributed's gloo backend move CUDA tensors point to point?  (two ranks on one device; decides whether a hang of
the bench's gather phase in the PTA_BENCH_BACKEND=gloo dry run says anything about the RCCL path)"""
import os, sys, time, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); torch.cuda.set_device(0)
dist.init_process_group("gloo")
t = torch.full((1 << 20,), float(rank + 1), device="cuda", dtype=torch.float64)
t0 = time.time()
try:
    if rank == 0:
        ops = [dist.P2POp(dist.irecv, t, 1)]
    else:
        ops = [dist.P2POp(dist.isend, t, 0)]
    works = dist.batch_isend_irecv(ops)
    for w in works: w.wait(timeout=__import__("datetime").timedelta(seconds=20))
    torch.cuda.synchronize()
    print(rank, "ok", float(t[0]), round(time.time() - t0, 2), flush=True)
except Exception as e:
    print(rank, "exception", type(e).__name__, str(e)[:200], flush=True)
os._exit(0)
