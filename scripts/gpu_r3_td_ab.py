"""TD product A/B on the headline array: kernel-only time of pta_td_trmm_rng (HIP events around the launch) with the deviates read from
memory and generated in registers, beside the whole generate_td()."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import _lib, device as dv
eng, psrs, noise = build_engine(68, 5000, seed=1)
eng.prepare_td()
R = 1024
out = dv.empty((R, eng.n_toa))
fl = float(sum(int(c) ** 2 for c in eng.counts))
s = dv.stream_ptr()
for name, draws in (("registers", "registers"), ("memory", "memory")):
    eng.td_draws = draws
    eng.generate_td(R, out=out); torch.cuda.synchronize()     # fills the plan (and Z)
    tp = eng.td_plan
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("pta_td_trmm_rng", ctypes.byref(tp), eng.seed, 0, R, dv.ptr(out), out.stride(0), s)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    tw = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.generate_td(R, out=out); torch.cuda.synchronize(); tw.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"variant": name, "product_kernel_ms": [round(t, 2) for t in ts], "useful_TFLOPs_kernel": round(fl * R / min(ts) / 1e9, 2),
                      "generate_td_ms": [round(t, 2) for t in tw]}), flush=True)
