"""generate_td at the headline size with the fp64 and the fp32 ("fast") Gaussian transform: how much of the L.z product's time is
the in-register deviate generation on the shared fp64 ALUs?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
eng = build_engine(68, 5000, 1)[0]
eng._gw = None
eng.prepare_td()
R = 1024
out = torch.empty((R, eng.n_toa), dtype=torch.float64, device="cuda")
res = {}
for name, fast in (("fp64_transform", 0), ("fp32_fast_transform", 1), ("fp64_transform_again", 0)):
    eng.rng_fast = fast
    eng.generate_td(R, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): eng.generate_td(R, out=out)
    b.record(); torch.cuda.synchronize()
    res[name + "_ms"] = round(a.elapsed_time(b) / 3, 2)
print(json.dumps(res))
