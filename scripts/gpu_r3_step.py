"""throughput-mode step A/B: fused kernel launch order (XCD-aware / linear) x white-noise draw (reference two deviates / single deviate), ms per 1024 realisations of the 68 x 5000 array."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import device as dv
eng, psrs, noise = build_engine(68, 5000, seed=20260921)
R = 1024
out = dv.empty((R, eng.n_toa))
def step_ms(K=20):
    for i in range(3): eng.generate(R, r0=i * R, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): eng.generate(R, r0=(3 + i) * R, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
res = {}
for rep in range(2):
    for sv in (0, 1):
        for wm in ("reference", "single"):
                eng.synth_variant, eng.wn_mode = sv, wm
                k = f"synth{sv}_{wm}"
                res[k] = min(res.get(k, 1e9), round(step_ms(), 4))
print(json.dumps(res))
