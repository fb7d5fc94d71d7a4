"""cProfile of the drop-in add_measurement_noise / add_jitter loops at 68 x 5000 (where do the 0.8 ms per call go?)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import headline_array
from pta_replicator_amd.simulate import make_ideal
from pta_replicator_amd.white_noise import add_measurement_noise, add_jitter
psrs, noise = headline_array(68, 5000)
def run():
    for p in psrs: make_ideal(p)
    for ii, p in enumerate(psrs): add_measurement_noise(p, efac=noise["efac"][ii], log10_equad=noise["log10_equad"][ii], flags=noise["flags"][ii], seed=10660 + ii)
    for ii, p in enumerate(psrs): add_jitter(p, log10_ecorr=noise["log10_ecorr"][ii], flags=noise["flags"][ii], coarsegrain=0.1, seed=17763 + ii)
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
