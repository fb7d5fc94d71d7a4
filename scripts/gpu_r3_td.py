"""TD-mode timing probe: prepare_td (assembly + factorisation) and generate_td(1024) on the headline array."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
from pta_replicator_amd import device as dv
eng, psrs, noise = build_engine(68, 5000, seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter(); eng.prepare_td(); torch.cuda.synchronize(); t_prep = time.perf_counter() - t0
R = 1024
out = dv.empty((R, eng.n_toa))
eng.generate_td(R, out=out); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.generate_td(R, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
fl = float(sum(int(c) ** 2 for c in eng.counts))
t = min(ts)
print(json.dumps({"prepare_td_ms": t_prep * 1e3, "generate_td_ms": [round(x * 1e3, 2) for x in ts], "realisations_per_s": R / t, "trmm_useful_TFLOPs": fl * R / t / 1e12}))
eng.td_draws = "memory"
eng.generate_td(R, out=out); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); eng.generate_td(R, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts)
print(json.dumps({"td_draws": "memory", "generate_td_ms": [round(x * 1e3, 2) for x in ts], "realisations_per_s": R / t, "trmm_useful_TFLOPs": fl * R / t / 1e12}))
