import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pta_replicator_amd.engine import ReplicaEngine
from pta_replicator_amd.simulate import ArrayTOAs, SimulatedPulsar, make_ideal

def build(components, seed=None):
    rng = np.random.default_rng(components if seed is None else seed)
    psrs = []
    for a, n in enumerate((777, 90, 1025, 2601)):
        ep = np.sort(rng.uniform(53000, 56000, n // 3 + 1))
        mjd = (ep[:, None] + rng.uniform(0, 0.01, (len(ep), 3))).ravel()[:n]
        p = SimulatedPulsar(toas=ArrayTOAs(mjd, rng.uniform(0.3, 1.5, n)), name=f"J{a:04d}", loc={"RAJ": 2.0 + 3 * a, "DECJ": -20.0 + 25 * a})
        make_ideal(p); psrs.append(p)
    eng = ReplicaEngine(psrs, seed=3)
    eng.set_white_noise(efac=1.1, log10_equad=-6.3)
    eng.set_jitter(log10_ecorr=-6.5, coarsegrain=0.1)
    eng.set_red_noise([-13.6, None, -14.0, -13.2], [3.1, None, 4.2, 2.2], components=components)
    eng.prepare()
    return eng

def attempt(eng, variant, mode, what):
    eng.td_cov_variant, eng.td_potrf_mode = variant, mode
    try:
        eng.prepare_td()
        r = "ok"
    except Exception as e:
        r = "FAILED " + str(e)[-50:]
    print(f"  {what}: variant {variant} mode {mode}: {r}", flush=True)

for comp, seed in ((29, 29), (29, 30), (30, 29), (30, 30), (32, 32)):
    eng = build(comp, seed)
    print("components", comp, "data seed", seed, "K", 2 * comp)
    for rep in range(2):
        for variant in (1, 2):
            for mode in ("ragged", "uniform"):
                attempt(eng, variant, mode, f"rep {rep}")
