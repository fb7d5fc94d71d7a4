"""generate_td(4096) on the 68 x 5000 array as four chunks of 1024: strictly sequential against the opt-in td_overlap (the deviates and the GWB
grid series of chunk c + 1 prepared on a side stream beside the product of chunk c)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json, bench, torch
from pta_replicator_amd import device as dv
eng, psrs, noise = bench.build_engine(68, 5000, seed=20260921)
eng.prepare_td()
R = 4096
out = dv.empty((R, eng.n_toa))
res = {}
flop = float(sum(int(n) ** 2 for n in eng.counts))
for name, ov in (("sequential_chunks_of_1024", False), ("prologue_of_next_chunk_beside_product", True), ("sequential_again", False), ("overlap_again", True)):
    eng.td_overlap, eng.td_chunk = ov, 1024
    eng._td_bufs = None
    eng.generate_td(R, out=out, chunk=1024)
    t = min(bench._wall(lambda: eng.generate_td(R, out=out, chunk=1024)) for _ in range(2))
    res[name] = {"ms_per_1024": t * 1e3 / 4, "frac": flop * R / t / 1e12 / 78.6}
ref = out.clone()
eng.td_overlap = False
eng.generate_td(R, out=out, chunk=1024)
res["bit_equal"] = bool(torch.equal(ref, out))
print(json.dumps(res))
