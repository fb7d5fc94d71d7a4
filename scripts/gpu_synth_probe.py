"""fused-kernel variants side by side (same box, same clocks)"""
import ctypes, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_engine
from pta_replicator_amd import _lib, device as dv

eng, _, _ = build_engine(68, 5000, 1)
R = 960
ws = eng.workspace(R)
out = dv.empty((R, eng.n_toa))
eng.generate(R, out=out)
s = dv.stream_ptr()

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

res, ref = {}, None
for var in (0, 1, 0, 1, 0, 1, 0, 1):
    _lib.call("pta_set_synth_variant", var)
    ms = timed(lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(out), out.stride(0), s))
    o = out[:16].clone()
    ref = o if ref is None else ref
    res.setdefault(f"v{var}_ms", []).append(round(ms, 4))
    res[f"v{var}_maxdiff"] = float((o - ref).abs().max())
_lib.call("pta_set_synth_variant", 0)
npts, P = eng.plan.gw_npts, eng.P
for mv in (0, 1, 0, 1):
    _lib.call("pta_set_mix_variant", mv)
    res.setdefault(f"mix{mv}_ms", []).append(round(timed(lambda: _lib.call("pta_gwb_mix", dv.ptr(eng.d_M), P, dv.ptr(ws["G0"]), R, npts, npts, dv.ptr(ws["G"]), s)), 4))
_lib.call("pta_set_mix_variant", 0)
print(json.dumps(res))
