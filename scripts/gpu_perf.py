"""Exploration harness for the GPU box: microbenchmarks and kernel-variant timings (not part of the product)."""
import ctypes
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pta_replicator_amd import _lib, device as dv
from bench import build_engine

res = ctypes.c_double(0.0)
out = {}
for bpc in (1, 2, 4, 8):
    _lib.call("pta_microbench", 0, bpc, 4000, ctypes.byref(res)); out[f"mfma_f64_tflops_w{bpc}"] = round(res.value, 2)
    _lib.call("pta_microbench", 1, bpc, 4000, ctypes.byref(res)); out[f"fma_f64_tflops_w{bpc}"] = round(res.value, 2)
    _lib.call("pta_microbench", 4, bpc, 400, ctypes.byref(res)); out[f"normals_T_w{bpc}"] = round(res.value, 4)
_lib.call("pta_microbench", 2, 1 << 31, 20, ctypes.byref(res)); out["hbm_write_TBps"] = round(res.value, 3)
_lib.call("pta_microbench", 3, 1 << 30, 20, ctypes.byref(res)); out["hbm_copy_TBps"] = round(res.value, 3)
print(json.dumps(out))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


R = int(sys.argv[1]) if len(sys.argv) > 1 else 960
_lib.call("pta_set_idft_variant", 1)
eng, psrs, noise = build_engine(68, 5000, seed=1)
s = dv.stream_ptr()
outb = dv.empty((R, eng.n_toa))
eng.generate(R, out=outb)
for w in (4, 6, 8):
    _lib.call("pta_set_synth_variant", w)
    t = timed(lambda: _lib.call("pta_engine_synth", ctypes.byref(eng.plan), eng.seed, 0, R, dv.ptr(outb), outb.stride(0), s))
    print(json.dumps({"synth_minw": w, "synth_ms": round(t, 3), "us_per_real": round(t * 1e3 / R, 3), "GBps": round(8.0 * eng.n_toa * R / t / 1e6, 1)}))
