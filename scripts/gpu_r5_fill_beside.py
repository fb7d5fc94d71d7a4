"""generate_td(1024) on the 68 x 5000 array: the main deviate fill on a second stream beside the GWB grid stage (td_fill_beside_gwb, opt-in)
against the default order - re-measured in round 5 because the grid stage no longer draws in registers (it is MFMA work now, the fill VALU + stores)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pta_replicator_amd import device as dv
eng, psrs, noise = bench.build_engine(68, 5000, seed=20260921)
eng.prepare_td()
out = dv.empty((1024, eng.n_toa))
res = {}
for name, v in (("default", False), ("fill_beside_gwb", True), ("default_again", False), ("fill_beside_gwb_again", True)):
    eng.td_fill_beside_gwb = v
    eng.generate_td(1024, out=out)
    res[name] = round(min(bench._wall(lambda: eng.generate_td(1024, out=out)) for _ in range(4)) * 1e3, 3)
print(json.dumps(res))
