"""PCIe-inclusive rate: generate R realisations and copy them to pinned host memory (double-buffered on two streams)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import device as dv

eng, _, _ = build_engine(68, 5000, 1)
R, steps = 480, 8
outs = [dv.empty((R, eng.n_toa)) for _ in range(2)]
hosts = [torch.empty((R, eng.n_toa), dtype=torch.float64).pin_memory() for _ in range(2)]
copy_stream = torch.cuda.Stream()
done = [torch.cuda.Event(), torch.cuda.Event()]
gen = [torch.cuda.Event(), torch.cuda.Event()]

def run(n):
    for i in range(n):
        b = i & 1
        torch.cuda.current_stream().wait_event(done[b])       # buffer b free again
        eng.generate(R, r0=i * R, out=outs[b])
        gen[b].record()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(gen[b])
            hosts[b].copy_(outs[b], non_blocking=True)
            done[b].record()
    torch.cuda.synchronize()

for e in done: e.record()
run(2)
t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
t1 = time.perf_counter()
for i in range(4):
    hosts[0].copy_(outs[0], non_blocking=True)
torch.cuda.synchronize(); dcopy = (time.perf_counter() - t1) / 4
print(json.dumps({"realisations_per_s_incl_d2h": R * steps / dt, "d2h_GBps": R * eng.n_toa * 8 / dcopy / 1e9,
                  "bytes_per_realisation": eng.n_toa * 8}))
