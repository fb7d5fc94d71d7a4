import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pta_replicator_amd import _lib, device as dv
from bench import build_engine
R = 960
eng, psrs, noise = build_engine(68, 5000, seed=1)
s = dv.stream_ptr()
outb = dv.empty((R, eng.n_toa)); eng.generate(R, out=outb)
ws = eng.workspace(R); P, Nf, npts = eng.P, eng.grid["Nf"], eng.plan.gw_npts
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = ctypes.c_double(0)
for cv in (0, 1):
    _lib.call("pta_set_czt_variant", cv)
    print(json.dumps({"czt_variant": cv, "ms": round(timed(lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, s)), 3)}))
_lib.call("pta_set_czt_variant", 0)
for var in (0, 6):
    _lib.call("pta_set_synth_variant", var)
    print(json.dumps({"synth_variant": var, "ms": round(timed(lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(outb), outb.stride(0), s)), 3)}))
_lib.call("pta_set_synth_variant", 0)
for fast in (0, 1):
    _lib.call("pta_set_rng_math", fast)
    t1 = timed(lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(outb), outb.stride(0), s))
    t2 = timed(lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, s))
    _lib.call("pta_microbench", 4, 8, 400, ctypes.byref(res))
    print(json.dumps({"fast": fast, "synth_ms": round(t1, 3), "czt_ms": round(t2, 3), "normals_T_per_s": round(res.value, 3)}))
_lib.call("pta_set_rng_math", 0)
# synth with signals switched off one at a time (which part costs what)
import copy
pl = eng.plan
saved = (pl.rn_k, pl.gw_npts, pl.wn_a, pl.ecorr_toa)
def t_synth():
    return round(timed(lambda: _lib.call("pta_engine_synth", ctypes.byref(pl), eng.seed, 0, R, dv.ptr(outb), outb.stride(0), s)), 3)
out = {"all": t_synth()}
pl.rn_k = 0; out["no_rn"] = t_synth(); pl.rn_k = saved[0]
pl.gw_npts = 0; out["no_gwb"] = t_synth(); pl.gw_npts = saved[1]
pl.wn_a = None; out["no_wn"] = t_synth(); pl.wn_a = saved[2]
pl.ecorr_toa = None; out["no_ecorr"] = t_synth(); pl.ecorr_toa = saved[3]
pl.rn_k = 0; pl.gw_npts = 0; pl.wn_a = None; pl.ecorr_toa = None; out["none(write only)"] = t_synth()
print(json.dumps(out))
