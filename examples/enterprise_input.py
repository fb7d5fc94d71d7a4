"""Enterprise-style pulsar objects as INPUT (BASELINE.json: "keeps the reference's injection API on enterprise Pulsar objects"): anything with
enterprise's array surface - name, toas [s], toaerrs [s], flags / backend_flags, a position - goes straight into the engine, or through
simulate.from_enterprise() into the add_* functions; realisations are handed back as enterprise-style objects.

    python examples/enterprise_input.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import configure_engine, headline_array
from pta_replicator_amd.engine import ReplicaEngine
from pta_replicator_amd.red_noise import add_gwb, add_red_noise
from pta_replicator_amd.simulate import from_enterprise

# stand-ins for enterprise.pulsar.Pulsar objects: the hand-off objects of this package (the reference produces the real ones, simulate.py:91-95)
psrs, noise = headline_array(8, 2000)
ent = [p.to_enterprise() for p in psrs]
print(type(ent[0]).__name__, ent[0].name, len(ent[0].toas), "TOAs, toas[0] =", ent[0].toas[0], "s")

# (1) the batched engine takes them as they are
eng = configure_engine(ReplicaEngine(ent, seed=7), noise)
rows = eng.generate(16)                                  # [16, sum N_toa] residuals on the GPU
print("engine from enterprise-style pulsars:", tuple(rows.shape), "rms", float(rows.square().mean().sqrt()) * 1e6, "us")

# (2) the drop-in API: wrap once, inject as usual, hand back
sim = [from_enterprise(e) for e in ent]
add_gwb(sim, log10_amplitude=-14.6, spectral_index=13.0 / 3.0, seed=1)
for i, p in enumerate(sim):
    add_red_noise(p, -13.8, 3.5, components=30, seed=100 + i)
back = [p.to_enterprise() for p in sim]
print("injected residual rms of", back[0].name, "=", float(np.sqrt(np.mean(back[0].residuals ** 2))) * 1e6, "us")
