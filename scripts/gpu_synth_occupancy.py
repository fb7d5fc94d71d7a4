"""fused synthesis kernel at the headline size against workgroups resident per CU: synth_variant 100 + k adds k KB of unused
dynamic LDS per workgroup (34 KB staging + k), so 4 / 3 / 2 / 1 workgroups fit the 160 KB of a CU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine
eng = build_engine(68, 5000, 1)[0]
R = 1024
out = torch.empty((R, eng.n_toa), dtype=torch.float64, device="cuda")
res = {}
for name, v in (("warm", 0), ("default_4wg", 0), ("pad12_3wg", 112), ("pad20_2wg", 120), ("pad60_1wg", 160), ("valu4", 4), ("valu8", 8)):
    eng.synth_variant = v
    eng.generate(R, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): eng.generate(R, out=out)
    b.record(); torch.cuda.synchronize()
    res[name] = round(a.elapsed_time(b) / 5, 3)
print(json.dumps({"ms_per_1024_realisations_whole_step": res}))
