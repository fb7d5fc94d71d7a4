"""Wall time of ONE realisation of the 68 x 5000 array through the drop-in add_* API (host-owned pulsars, NumPy legacy draws on the
host in the reference's order, PCIe transfers included): the reference-style loop of per-pulsar calls and the list form."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import api_mode_timing, headline_array
psrs, noise = headline_array(68, 5000)
print(json.dumps(api_mode_timing(psrs, noise)))
if "--profile" in sys.argv:   # where the list form's host time goes (cProfile, cumulative)
    import cProfile, pstats
    from pta_replicator_amd.simulate import make_ideal
    from pta_replicator_amd.white_noise import add_measurement_noise, add_jitter
    from pta_replicator_amd.red_noise import add_red_noise, add_gwb
    P = len(psrs)

    def one():
        add_gwb(psrs, noise["gw_log10_A"], 13. / 3., seed=16672)
        add_measurement_noise(psrs, efac=noise["efac"], log10_equad=noise["log10_equad"], flags=noise["flags"], seed=[10660 + i for i in range(P)])
        add_jitter(psrs, log10_ecorr=noise["log10_ecorr"], flags=noise["flags"], coarsegrain=0.1, seed=[17763 + i for i in range(P)])
        add_red_noise(psrs, noise["rn_log10_A"], noise["rn_gamma"], components=30, seed=[19870 + i for i in range(P)])
        torch.cuda.synchronize()
    for p in psrs:
        make_ideal(p)
    one()
    for p in psrs:
        make_ideal(p)
    pr = cProfile.Profile()
    pr.enable()
    one()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
