"""Wall time of ONE realisation through the drop-in API (host-owned pulsars, PCIe transfers included) at 68 x 5000."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import headline_array
from pta_replicator_amd.simulate import make_ideal
from pta_replicator_amd.white_noise import add_measurement_noise, add_jitter
from pta_replicator_amd.red_noise import add_red_noise, add_gwb
psrs, noise = headline_array(68, 5000)
def one():
    for p in psrs: make_ideal(p)
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    add_gwb(psrs, noise["gw_log10_A"], 13. / 3., seed=16672); torch.cuda.synchronize(); t["gwb"] = time.perf_counter() - t0; t0 = time.perf_counter()
    for ii, p in enumerate(psrs): add_measurement_noise(p, efac=noise["efac"][ii], log10_equad=noise["log10_equad"][ii], flags=noise["flags"][ii], seed=10660 + ii)
    torch.cuda.synchronize(); t["wn"] = time.perf_counter() - t0; t0 = time.perf_counter()
    for ii, p in enumerate(psrs): add_jitter(p, log10_ecorr=noise["log10_ecorr"][ii], flags=noise["flags"][ii], coarsegrain=0.1, seed=17763 + ii)
    torch.cuda.synchronize(); t["ecorr"] = time.perf_counter() - t0; t0 = time.perf_counter()
    for ii, p in enumerate(psrs):
        if noise["rn_log10_A"][ii] is not None: add_red_noise(p, noise["rn_log10_A"][ii], noise["rn_gamma"][ii], components=30, seed=19870 + ii)
    torch.cuda.synchronize(); t["rn"] = time.perf_counter() - t0
    t["total"] = sum(t.values()); return t
one()
print(json.dumps({k: round(v, 4) for k, v in one().items()}))
