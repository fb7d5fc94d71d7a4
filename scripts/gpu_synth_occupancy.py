"""fused synthesis kernel at the headline size against workgroups resident per CU: synth_variant 100 + k adds k KB of unused
dynamic LDS per workgroup (37 KB staging + tables + k), so 4 / 3 / 2 / 1 workgroups fit the 160 KB of a CU.  Every variant is
timed in 4 interleaved rounds of 8 steps; the minimum and the median are reported (clock ramps make single readings noisy by
several percent).  Also times the synthesis kernel alone (pta_engine_synth on the plan of the last generate())."""
import ctypes, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import _lib, device as dv
eng = build_engine(68, 5000, 1)[0]
R = 1024
out = torch.empty((R, eng.n_toa), dtype=torch.float64, device="cuda")
variants = (("default_4wg", 0), ("pad12_3wg", 112), ("pad20_2wg", 120), ("pad60_1wg", 160), ("valu4", 4))
def timed(fn, n=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
step = {k: [] for k, _ in variants}
kern = {k: [] for k, _ in variants}
for _ in range(3): eng.generate(R, out=out)
s = dv.stream_ptr()
for rnd in range(4):
    for name, v in variants:
        eng.synth_variant = v
        eng.generate(R, out=out)
        step[name].append(timed(lambda: eng.generate(R, out=out)))
        kern[name].append(timed(lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(out), out.stride(0), s)))
fmt = lambda d: {k: [round(min(v), 3), round(statistics.median(v), 3)] for k, v in d.items()}
print(json.dumps({"step_ms_min_median": fmt(step), "synth_kernel_ms_min_median": fmt(kern)}))
