"""Where does the chirp-z kernel's time go?  FFT-only (draws broadcast from one cached row) vs full (draws on chip)."""
import ctypes, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_engine
from pta_replicator_amd import _lib, device as dv

eng, _, _ = build_engine(68, 5000, 1)
R, P, Nf, npts = 960, eng.P, eng.grid["Nf"], eng.plan.gw_npts
ws = eng.workspace(R)
s = dv.stream_ptr()
w = dv.f64(np.random.default_rng(0).standard_normal(2 * Nf))

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out = {}
ref = None
for var in (0, 1, 0, 1, 25):
    try:
        _lib.call("pta_set_czt_variant", var)
        out[f"v{var}_full_ms"] = timed(lambda: _lib.call("pta_gwb_czt", eng.seed, 0, None, 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, s))
        g = ws["G0"][:4 * P].clone()
        ref = g if ref is None else ref
        out[f"v{var}_maxdiff_rel"] = float((g - ref).abs().max() / ref.abs().max())
        out[f"v{var}_fft_only_ms"] = timed(lambda: _lib.call("pta_gwb_czt", eng.seed, 0, dv.ptr(w), 0, R, P, Nf, npts, 10, *[dv.ptr(x) for x in eng.d_czt], dv.ptr(ws["G0"]), npts, s))
    except Exception as e:
        out[f"v{var}_error"] = str(e)
_lib.call("pta_set_czt_variant", 0)
res = ctypes.c_double(0.0)
_lib.call("pta_microbench", 4, 8, 200, ctypes.byref(res))
out["normals_T_per_s"] = res.value
out["rng_only_ms_estimate"] = R * P * 2 * (Nf - 2) / (res.value * 1e12) * 1e3
print(json.dumps(out))
